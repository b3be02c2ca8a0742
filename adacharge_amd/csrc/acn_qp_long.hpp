// Long-horizon kernel of the batched MPC QP solver for gfx950: horizons of 33 ... 288 periods (the reference's
// N = 54 x T = 144 stress scenarios, tests/test_adacharge_stress.py; day-long offline problems, aco.py:403-408),
// any site of up to 32 padded rows -- the shapes neither the register-resident kernel (T <= 32) nor the large-site
// kernel (T <= 48, whole rows of every tile in registers AND LDS slabs sized by the horizon) can hold.
//
// Same ADMM as acn_qp_tiled.hpp / acn_qp_stream.hpp.  What changes is who owns what.  Only ONE step of the iteration
// couples the periods of an EVSE: the projection onto its energy rows (water-filling over the session window).
// Everything else -- P = Ghat r0, the eigen-space step, the site-row projection, x~ = (r0 + Ghat' e^) / a -- is
// independent per period.  So an iteration is two phases over two kinds of work item, one wave per item:
//
//   phase 1, item = one COLUMN TILE (16 periods, every EVSE tile and every site-row tile of it):
//     P[:, c] = sum_e Ghat[:, e] r0[e, c]  ->  e^, h^  ->  G x~ = Q h^, relaxation, projection onto C, y2
//     ->  x~[e, c] = (r0[e, c] + Ghat[:, e]' e^) / a for every e, relaxation, zh[e, c] -> workspace.
//     The whole chain runs in the wave's registers: an MFMA accumulator tile IS the B operand of the next product
//     (the site fragments are stored pre-permuted for that), so no LDS and no barrier inside the phase.
//   phase 2, item = one REGISTER ROW of an EVSE tile (4 EVSEs x the WHOLE horizon, CTL column registers per lane):
//     zh, lb, ub -> water-filling (safeguarded Newton along the 16-lane DPP rows) -> z1, y1, the new r0 -> workspace.
//
// Two barriers per iteration, whatever the horizon.  With 16 waves per problem a 54 x 144 problem has 9 column items
// and 16 row items: every phase is one step deep, and the latency of an iteration is a handful of dependent L2
// round trips.  State streams through a per-problem workspace in MFMA fragment order (L2 / MALL resident at these
// sizes: 54 x 144 is 74 KB per array); the site-row state of a column is only ever touched by the wave that owns it.
//
// Anderson acceleration as in the other kernels (the ring lives in the workspace); the infeasibility certificate of
// acn_qp_tiled.hpp.  Not in this kernel (the general-shape kernel keeps it): the demand-charge row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acn_qp_stream.hpp"

namespace acnqp {

constexpr int kLongAccelMax = 5;   // Anderson columns (the ring lives in the workspace: any shape takes all five)

// doubles of workspace one problem needs (accel = Anderson columns in use)
__host__ __device__ inline long long long_workspace(int NP, int CTL, int K, int MT, int accel) {
  const long long NT = (long long)(NP / 16) * CTL * 256, MS = (long long)MT * CTL * 256;
  long long w = 7 * NT + (long long)K * NP + 3 * MS + (NT + MS + 1) / 2 + 64;   // + the certificate's dual snapshot (floats)
  if (accel > 0) w += MS + 3 * (NT + MS) + (2LL * accel * (NT + MS) * 4 + 7) / 8;   // zhr; u, f, g; the float rings
  return w;
}

// CTL: column registers of a row item (>= ceil(Tm / 16)); NWV: waves per problem (16: 128 registers per lane, rows of
// up to 9 column tiles; 8: 256 registers, rows of up to 18)
template <int CTL, int MT, int NWV>
__global__ __launch_bounds__(NWV * 64, 1) void admm_long_kernel(const StreamArgs SA) {
  using M = Mfma<double>;
  using vec4 = M::vec4;
  typedef double real;
  const TiledArgs& A = SA.t;
  constexpr int AMX = kLongAccelMax;
  __shared__ real SC[NWV * 8 + 8];
  __shared__ real AaRedS[NWV * (AMX + 2)];
  __shared__ real AaHS[NWV * (AMX * AMX + AMX)];

  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
  int g = lane >> 4, t = lane & 15;
#define RELANE() do { asm volatile("" : "+v"(lane)); g = lane >> 4; t = lane & 15; } while (0)
  const int N = A.N, Tm = A.Tm, NP = A.NP, K = A.K;
  const int NE = NP >> 4;                 // EVSE tiles
  const int nct = (Tm + 15) >> 4;         // column tiles that hold periods (<= CTL); the rest is padding, never touched
  const long long NT = (long long)NE * CTL * 256;
  real* W0 = SA.work + (size_t)b * SA.ws_per_problem;
  real *Xs = W0, *Z1s = Xs + NT, *Y1s = Z1s + NT, *Qs = Y1s + NT, *LBs = Qs + NT, *UBs = LBs + NT;
  real* RZ = UBs + NT;                    // r0 (written by phase 2, read by phase 1) / zh (the other way round)
  real* MU = RZ + NT;                     // [K][NP]
  real* Z2 = MU + (size_t)K * NP;         // site-row state in tile-fragment order [MT][CTL][4][64]
  real* Y2 = Z2 + MT * CTL * 256;
  real* GX = Y2 + MT * CTL * 256;
  // Anderson acceleration (acn_qp_tiled.hpp, oracle/admm_port.c): pre-projection site rows of an event iteration, the
  // previous event's u / f / g over the EVSE part [0, NT) and the site part [NT, NT + MS), the dF / dG rings (floats)
  const int aa_m = min(max(A.accel_mem, 0), AMX);
  const unsigned MS = (unsigned)(MT * CTL * 256), DU = (unsigned)NT + MS;
  float* Y1P = reinterpret_cast<float*>(GX + MS);   // duals at the previous check (infeasibility certificate)
  float* Y2P = Y1P + NT;
  real* ZHR = GX + MS + (NT + MS + 1) / 2;
  real* UP = ZHR + MS;
  real* FP = UP + DU;
  real* GP = FP + DU;
  float* HF = reinterpret_cast<float*>(GP + DU);
  float* HG = HF + (size_t)aa_m * DU;
  real* AaH = AaHS + wave * (AMX * AMX + AMX);   // this wave's copy of (H, b): every wave runs the small solve itself
  const real* FG = static_cast<const real*>(A.fragG);
  const real* FQ = static_cast<const real*>(A.fragQ);
  const real* Gm = static_cast<const real*>(A.G);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);
  const bool eq = A.s_eq[b] != 0;
  const real sigma = A.sigma, alpha = A.alpha;
  const real lfb = A.lf ? A.lf[b] / (A.flat_scale * A.flat_scale) : 0.0;

  // offsets inside one problem's arrays fit 32 bits: scalar base + per-lane offset addressing
  auto fidx = [&](int e, int c, int r) -> unsigned { return (unsigned)(((e * CTL + c) * 4 + r) * 64 + lane); };
  auto sidx2 = [&](int m, int c, int r) -> unsigned { return (unsigned)(((m * CTL + c) * 4 + r) * 64 + lane); };

  // ---- init (row items): inputs -> fragment order; |q|_inf, max ub; a session whose bounds cannot meet its energy
  // row ----------------------------------------------------------------------------------------------------------
  real qn = 0, um = 0, bad = 0;
#pragma unroll 1
  for (int ri = wave; ri < 4 * NE; ri += NWV) {
    RELANE();
    const int e = ri >> 2, r = ri & 3;
    const int ev = 16 * e + M::rowof(g, r);
    real lbv[CTL], ubv[CTL];
#pragma unroll
    for (int c = 0; c < CTL; ++c) {
      lbv[c] = 0; ubv[c] = 0;
      if (c < nct) {
        const int tt = 16 * c + t;
        const bool ok = ev < N && tt < Tm;
        const size_t idx = ((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0);
        const real l = ok ? A.lb[idx] : 0.0;
        real u = ok ? A.ub[idx] : 0.0;
        const real q = ok ? A.q[idx] : 0.0;
        if (u < l) u = l;
        lbv[c] = l; ubv[c] = u;
        const unsigned i = fidx(e, c, r);
        LBs[i] = l; UBs[i] = u; Qs[i] = q;
        qn = fmax(qn, fabs(q)); um = fmax(um, u);
      }
    }
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const size_t sidx = ((size_t)b * K + k) * N + (ev < N ? ev : 0);
      const int off = ev < N ? A.s_off[sidx] : 0, len = ev < N ? A.s_len[sidx] : 0;
      real sl = 0, su = 0;
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const int tp = 16 * c + t;
        const bool inw = tp >= off && tp < off + len && tp < Tm;
        sl += inw ? lbv[c] : 0.0; su += inw ? ubv[c] : 0.0;
      }
      sl = row_sum<real>(sl); su = row_sum<real>(su);
      if (len > 0) {
        const real cap = A.s_cap[sidx];
        const real slack = 64.0 * M::proj_tol * fmax(1.0, fabs(cap));
        if (sl > cap + slack || (eq && su < cap - slack)) bad = 1;
      }
      if (t == 0 && ev < NP) MU[(size_t)k * NP + ev] = 0;
    }
  }
  real qnorm, pd;
  const real pd_user = A.pdiag[b];
  {
    real f[3] = {qn, um, bad};
    stream_block_max<3, NWV>(f, SC, lane, wave);
    qnorm = f[0];
    pd = effective_pdiag<real>(pd_user, A.reg_rel, qnorm, f[1], A.horizon[b], lfb > 0.0);
    if (f[2] > 0) {
      for (size_t k = tid; k < (size_t)N * Tm; k += NWV * 64) A.x[(size_t)b * N * Tm + k] = 0;
      if (tid == 0) { A.status[b] = 4; A.iters[b] = 0; A.pri[b] = M::big; A.dua[b] = M::big; A.obj[b] = 0; }
      return;
    }
  }

  real rho = A.rho0;

  // ---- projection of ONE register row (EVSE 16 e + rowof(g, r): its periods are the 16 lanes of a DPP row times the
  // CTL column registers) onto B = box + energy rows: the safeguarded Newton of the other kernels / the C port ------
  auto project_row = [&](int e, int r, const real (&zh)[CTL], const real (&lbv)[CTL], const real (&ubv)[CTL],
                         real (&z1)[CTL], bool reset_mu) __attribute__((always_inline)) {
    const int ev = 16 * e + M::rowof(g, r);
#pragma unroll
    for (int c = 0; c < CTL; ++c) z1[c] = fmin(fmax(zh[c], lbv[c]), ubv[c]);
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const size_t sidx = ((size_t)b * K + k) * N + (ev < N ? ev : 0);
      const int off = ev < N ? A.s_off[sidx] : 0;
      int len = ev < N ? A.s_len[sidx] : 0;
      if (off + len > Tm) len = Tm - off;
      const real cap = ev < N ? A.s_cap[sidx] : 0.0;
      real s0 = 0, sl = 0, su = 0, lo_l = M::big, hi_l = -M::big;
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const int tp = 16 * c + t;
        const bool inw = tp >= off && tp < off + len;
        s0 += inw ? z1[c] : 0.0;
        sl += inw ? lbv[c] : 0.0;
        su += inw ? ubv[c] : 0.0;
        lo_l = inw ? fmin(lo_l, zh[c] - ubv[c]) : lo_l;
        hi_l = inw ? fmax(hi_l, zh[c] - lbv[c]) : hi_l;
      }
      s0 = row_sum<real>(s0); sl = row_sum<real>(sl); su = row_sum<real>(su);
      real lo = row_min<real>(lo_l), hi = row_max<real>(hi_l);
      const real tol = M::proj_tol * fmax(1.0, fabs(cap));
      const bool act = len > 0 && (eq ? fabs(s0 - cap) > tol : s0 > cap + tol);
      const int mode = !act ? 4 : ((eq && cap >= su) ? 2 : (cap <= sl ? 3 : 0));   // 0 root-find, 2 at ub, 3 at lb, 4 nothing
      bool need = mode == 0;
      if (!eq && lo < 0) lo = 0;
      const real mu0 = (reset_mu || ev >= N) ? 0.0 : MU[(size_t)k * NP + ev];
      real m = fmin(fmax(mu0, lo), hi);
#pragma unroll 1
      for (int guard = 0; guard <= 100; ++guard) {
        if (!__any(need)) break;
        real gl = 0, nl = 0;
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          const int tp = 16 * c + t;
          const bool inw = tp >= off && tp < off + len;
          const real u = zh[c] - m;
          gl += inw ? fmin(fmax(u, lbv[c]), ubv[c]) : 0.0;
          nl += (inw && u > lbv[c] && u < ubv[c]) ? 1.0 : 0.0;
        }
        const real gs = row_sum<real>(gl), nf = row_sum<real>(nl);
        const real d = gs - cap;
        need = need && !(fabs(d) <= tol);
        lo = (need && d > 0) ? m : lo;
        hi = (need && !(d > 0)) ? m : hi;
        real mn = nf > 0 ? m + d / nf : 0.5 * (lo + hi);
        if (!(mn > lo && mn < hi)) mn = 0.5 * (lo + hi);
        m = need ? mn : m;
      }
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const int tp = 16 * c + t;
        if (tp >= off && tp < off + len) {
          if (mode == 0) z1[c] = fmin(fmax(zh[c] - m, lbv[c]), ubv[c]);
          else if (mode == 2) z1[c] = ubv[c];
          else if (mode == 3) z1[c] = lbv[c];
        }
      }
      if (t == 0 && ev < N) MU[(size_t)k * NP + ev] = (mode == 0 && !reset_mu) ? m : 0.0;
    }
  };

  // P[:, c] = sum over EVSE tiles of Ghat[:, e] v[e, c], v = the RZ array (r0, or z1 during the start), in tile order.
  // The operands of a tile are requested as one batch, a tile ahead of the MFMAs that consume them (two register
  // sets): left to itself the compiler issues one load per MFMA and waits for each.
  auto load_p = [&](int e, int c, const real* fgb, real (&bv)[4], real (&af)[MT][4]) __attribute__((always_inline)) {
    const real* fg = fgb + (size_t)e * MT * 2 * 4 * 64 + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = RZ[fidx(e, c, s)];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s = 0; s < 4; ++s) af[m][s] = fg[((m * 2 + 0) * 4 + s) * 64];
  };
  auto mma_p = [&](const real (&bv)[4], const real (&af)[MT][4], vec4 (&p)[MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MT; ++m) p[m] = M::mma(af[m][s], bv[s], p[m]);
  };
  auto column_p = [&](int c, const real* fgb, vec4 (&p)[MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m) p[m] = vec4{0, 0, 0, 0};
    real bvA[4], afA[MT][4], bvB[4], afB[MT][4];
    load_p(0, c, fgb, bvA, afA);
#pragma unroll 1
    for (int e = 0; e < NE; e += 2) {
      if (e + 1 < NE) load_p(e + 1, c, fgb, bvB, afB);
      __builtin_amdgcn_sched_barrier(0);
      mma_p(bvA, afA, p);
      __builtin_amdgcn_sched_barrier(0);
      if (e + 2 < NE) load_p(e + 2, c, fgb, bvA, afA);
      __builtin_amdgcn_sched_barrier(0);
      if (e + 1 < NE) mma_p(bvB, afB, p);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // r0 of column c from the stored state (start, and after a rho change): column items, own columns only
  auto rebuild_r0 = [&](int c) __attribute__((always_inline)) {
#pragma unroll 2
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned i = fidx(e, c, r);
        RZ[i] = sigma * Xs[i] - Qs[i] + rho * Z1s[i] - Y1s[i];
      }
  };

  // u = (z1 + y1 / rho, z2 + y2 / rho) of column c: the Anderson state at the start and after a rho change
  auto reset_u = [&](int c) __attribute__((always_inline)) {
    const real ir = 1.0 / rho;
#pragma unroll 2
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const unsigned i = fidx(e, c, r); UP[i] = Z1s[i] + Y1s[i] * ir; }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(m, c, r); UP[(unsigned)NT + i] = Z2[i] + Y2[i] * ir; }
  };
  // projection of one site-row tile onto C from its pre-projection point: z2, y2, and the tile's residual terms
  real sv0 = 0, sv2 = 0;   // |G x - z2|_inf, max(|G x|, |z2|): the site-row share of the residuals, per iteration
  auto site_project = [&](int mo, int c, const real (&zhr)[4], const real (&gxn)[4], const int (&ty)[4],
                          const real (&lim)[4], real pk) __attribute__((always_inline)) {
    real scl[2] = {1.0, 1.0};
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
      if (ty[2 * pr] == kRowSocRe) {
        const real re = zhr[2 * pr], im = zhr[2 * pr + 1];
        const real n2 = re * re + im * im;
        if (n2 > lim[2 * pr] * lim[2 * pr]) scl[pr] = lim[2 * pr] / sqrt(n2);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned i = sidx2(mo, c, r);
      real zn = zhr[r];
      if (ty[r] == kRowBox) zn = fmin(zn, lim[r]);
      else if (ty[r] == kRowPeak) zn = fmin(zn, pk);
      else if (ty[r] == kRowQuad) zn = zn * (rho / (rho + lfb));
      else if (ty[r] == kRowSocRe || ty[r] == kRowSocIm) zn = zn * scl[r >> 1];
      Y2[i] = rho * (zhr[r] - zn);
      Z2[i] = zn;
      sv0 = fmax(sv0, fabs(gxn[r] - zn));
      sv2 = fmax(sv2, fmax(fabs(gxn[r]), fabs(zn)));
    }
  };

  // ---- start (see acn_qp_tiled.hpp): z1 = Proj_B(-kStartGain q), x = z1, y1 = -(q + pd z1); z2 = G z1 = Q (Ghat z1);
  // warm (optional): z1 = Proj_B(warm_x), y2 = warm_y, y1 = -(q + pd z1 + G' y2) -------------------------------------
  const bool warm = A.warm_x != nullptr && A.warm_y != nullptr;
#pragma unroll 1
  for (int ri = wave; ri < 4 * NE; ri += NWV) {
    RELANE();
    const int e = ri >> 2, r = ri & 3;
    real zs[CTL], lbv[CTL], ubv[CTL], z1[CTL];
#pragma unroll
    for (int c = 0; c < CTL; ++c) {
      zs[c] = 0; lbv[c] = 0; ubv[c] = 0;
      if (c < nct) {
        const unsigned i = fidx(e, c, r);
        lbv[c] = LBs[i]; ubv[c] = UBs[i];
        zs[c] = -kStartGain * Qs[i];
        if (warm) {
          const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
          const bool ok = ev < N && tt < Tm;
          zs[c] = ok ? A.warm_x[((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0)] : 0.0;
        }
      }
    }
    project_row(e, r, zs, lbv, ubv, z1, true);
#pragma unroll
    for (int c = 0; c < CTL; ++c)
      if (c < nct) {
        const unsigned i = fidx(e, c, r);
        Xs[i] = z1[c]; Z1s[i] = z1[c]; Y1s[i] = -(Qs[i] + pd * z1[c]); RZ[i] = z1[c];
      }
  }
  __syncthreads();
#pragma unroll 1
  for (int c = wave; c < nct; c += NWV) {
    RELANE();
    vec4 p[MT];
    column_p(c, FG, p);
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      vec4 zt = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          zt = M::mma(FQ[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], p[mi][s], zt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned i = sidx2(mo, c, r);
        real yv = 0;
        if (warm) {
          const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
          const int ja = A.rowabi[j];
          if (ja >= 0 && tt < Tm) yv = A.warm_y[((size_t)b * A.Mg + ja) * Tm + tt] / static_cast<const real*>(A.rowscale)[j];
        }
        Z2[i] = zt[r]; GX[i] = zt[r]; Y2[i] = yv;
      }
    }
    if (warm) {   // y1 = -(q + pd z1 + G' y2): the row items stored the G' y2 = 0 version
#pragma unroll 1
      for (int e = 0; e < NE; ++e) {
        vec4 gty = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[sidx2(m, c, s)], gty);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y1s[fidx(e, c, r)] -= gty[r];
      }
    }
    rebuild_r0(c);
    if (aa_m > 0) reset_u(c);
  }
  // the main loop's phase 1 reads what the same wave just wrote (columns keep their owner): no barrier

  int status = 2, it = 0, n_adapt = 0;
  real pri = M::big, dua = M::big;
  bool done = false, have_prev = false;
  // Anderson state (block-uniform scalars, every wave keeps its own identical copy)
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  if (aa_m > 0)
    for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
#ifdef ACNQP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
#pragma unroll 1
  while (!done) {
    ++it;
    const real a = sigma + pd + rho, inv_a = 1.0 / a, inv_rho = 1.0 / rho;
    const bool check = (it % A.check_every == 0) || it >= A.max_iter;
    // an offset the compiler cannot see through keeps the loads of loop-invariant site data inside the loop (L1 / L2
    // hits) instead of pinning registers across it
    unsigned zoff = 0;
    asm volatile("" : "+s"(zoff));
    const real* FQi = FQ + zoff;
    const real* FGi = FG + zoff;
    const real* Lmi = Lm + zoff;
    const real* RLi = RL + zoff;
    const int32_t* RTi = A.rowtype + zoff;
    sv0 = 0; sv2 = 0;
    const bool ev_it = aa_m > 0 && it % kAaPeriod == 0;   // Anderson event: the site rows are projected after it
    // ================= phase 1: column items ===================================================================
#pragma unroll 1
    for (int c = wave; c < nct; c += NWV) {
      RELANE();
      // the column's site-row state and the fragments of the eigen step: requested now, used after P
      real z2v[MT][4], y2v[MT][4], fq0[MT][MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(m, c, r); z2v[m][r] = Z2[i]; y2v[m][r] = Y2[i]; }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) fq0[mo][mi][s] = FQi[(((mo * MT + mi) * 2 + 0) * 4 + s) * 64 + lane];
      vec4 p[MT];
      column_p(c, FGi, p);
      STAMP(0);   // P of the column
      // requested now, used after the eigen step: the fragments of Q h^, G x, the row constants
      real fq1[MT][MT][4], gxv[MT][4], ljv[MT][4], lim[MT][4];
      int ty[MT][4];
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) fq1[mo][mi][s] = FQi[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * mo + M::rowof(g, r);
          gxv[mo][r] = GX[sidx2(mo, c, r)];
          ljv[mo][r] = Lmi[j]; lim[mo][r] = RLi[j]; ty[mo][r] = RTi[j];
        }
      }
      const int tt = 16 * c + t;
      real pk = M::big;
      if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
      __builtin_amdgcn_sched_barrier(0);
      // ---- eigen space: e^, h^ (accumulator layout = the B operand of the products below) ----------------------
      vec4 eh[MT], hh[MT];
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        vec4 wh = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) wh = M::mma(fq0[mo][mi][s], rho * z2v[mi][s] - y2v[mi][s], wh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real lj = ljv[mo][r];
          const real e_ = wh[r] - (rho / (a + rho * lj)) * (p[mo][r] + lj * wh[r]);
          eh[mo][r] = e_;
          hh[mo][r] = (p[mo][r] + lj * e_) * inv_a;
        }
      }
      // first EVSE tile of the x~ loop below: requested before the site-row products
      real rA[4], xA[4], zA[4], yA[4], fA[MT][4], rB[4], xB[4], zB[4], yB[4], fB[MT][4];
      auto load_x = [&](int e, real (&rv)[4], real (&xv)[4], real (&zv)[4], real (&yv)[4], real (&fx)[MT][4]) __attribute__((always_inline)) {
        const real* fg = FGi + (size_t)e * MT * 2 * 4 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned i = fidx(e, c, r);
          rv[r] = RZ[i]; xv[r] = Xs[i]; zv[r] = Z1s[i]; yv[r] = Y1s[i];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) fx[m][s] = fg[((m * 2 + 1) * 4 + s) * 64];
      };
      load_x(0, rA, xA, zA, yA, fA);
      __builtin_amdgcn_sched_barrier(0);
      // ---- site rows: G x~ = Q h^, relaxation, projection onto C, y2 ---------------------------------------------
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        vec4 zt = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) zt = M::mma(fq1[mo][mi][s], hh[mi][s], zt);
        real zhr[4], gxn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gxn[r] = alpha * zt[r] + (1.0 - alpha) * gxv[mo][r];
          zhr[r] = alpha * zt[r] + (1.0 - alpha) * z2v[mo][r] + y2v[mo][r] * inv_rho;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) GX[sidx2(mo, c, r)] = gxn[r];
        if (ev_it) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ZHR[sidx2(mo, c, r)] = zhr[r];
        } else {
          site_project(mo, c, zhr, gxn, ty[mo], lim[mo], pk);
        }
      }
      STAMP(1);   // eigen step, site rows
      // ---- x~ of every EVSE tile of the column; zh takes r0's place (two register sets, as for P) -----------------
      auto do_x = [&](int e, const real (&rv)[4], const real (&xv)[4], const real (&zv)[4], const real (&yv)[4],
                      const real (&fx)[MT][4]) __attribute__((always_inline)) {
        vec4 acc = {rv[0], rv[1], rv[2], rv[3]};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = M::mma(fx[m][s], eh[m][s], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned i = fidx(e, c, r);
          const real xn = acc[r] * inv_a;
          RZ[i] = alpha * xn + (1.0 - alpha) * zv[r] + yv[r] * inv_rho;
          Xs[i] = alpha * xn + (1.0 - alpha) * xv[r];
        }
      };
#pragma unroll 1
      for (int e = 0; e < NE; e += 2) {
        if (e + 1 < NE) load_x(e + 1, rB, xB, zB, yB, fB);
        __builtin_amdgcn_sched_barrier(0);
        do_x(e, rA, xA, zA, yA, fA);
        __builtin_amdgcn_sched_barrier(0);
        if (e + 2 < NE) load_x(e + 2, rA, xA, zA, yA, fA);
        __builtin_amdgcn_sched_barrier(0);
        if (e + 1 < NE) do_x(e + 1, rB, xB, zB, yB, fB);
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP(2);   // x~, zh
    }
    STAMP(2);
    if (ev_it) {
      // ---- Anderson event (type II, acn_qp_tiled.hpp / oracle/admm_port.c): u = (zh, zhr) is the state of the
      // fixed-point map.  Column items again: the wave that wrote a column's zh / zhr reads them back.
      const bool col = aa_have_prev;
      const int slot = aa_head;
      real d[AMX + 2];
#pragma unroll
      for (int j = 0; j < AMX + 2; ++j) d[j] = 0;
      // four registers of one tile: g = zsrc[zo + 64 r], state index uo + 64 r
      auto aa_tile = [&](const real* zsrc, unsigned zo, unsigned uo) __attribute__((always_inline)) {
        real gv[4], uv[4], fpv[4], gpv[4];
        float hv[AMX][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { gv[r] = zsrc[zo + 64 * r]; uv[r] = UP[uo + 64 * r]; fpv[r] = FP[uo + 64 * r]; gpv[r] = GP[uo + 64 * r]; }
#pragma unroll
        for (int j = 0; j < AMX; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hv[j][r] = 0.f;
            if (((aa_valid >> j) & 1u) && j != slot) hv[j][r] = HF[(size_t)j * DU + uo + 64 * r];   // uniform
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real f = gv[r] - uv[r];
          d[AMX + 1] += f * f;
          const float cq = (float)(f - fpv[r]);
          if (col) { HF[(size_t)slot * DU + uo + 64 * r] = cq; HG[(size_t)slot * DU + uo + 64 * r] = (float)(gv[r] - gpv[r]); }
#pragma unroll
          for (int j = 0; j < AMX; ++j) d[j] += (real)cq * (real)(j == slot ? cq : hv[j][r]);
          d[AMX] += (real)cq * f;
          FP[uo + 64 * r] = f; GP[uo + 64 * r] = gv[r];
        }
      };
#pragma unroll 1
      for (int c = wave; c < nct; c += NWV) {
        RELANE();
#pragma unroll 1
        for (int e = 0; e < NE; ++e) aa_tile(RZ, fidx(e, c, 0), fidx(e, c, 0));
#pragma unroll
        for (int m = 0; m < MT; ++m) aa_tile(ZHR, sidx2(m, c, 0), (unsigned)NT + sidx2(m, c, 0));
      }
#pragma unroll
      for (int j = 0; j < AMX + 2; ++j) d[j] = wave_sum<real>(d[j]);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < AMX + 2; ++j) AaRedS[wave * (AMX + 2) + j] = d[j];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < AMX + 2; ++j) {
        real sw = 0;
        for (int wv = 0; wv < NWV; ++wv) sw += AaRedS[wv * (AMX + 2) + j];
        d[j] = sw;
      }
      const real fn = sqrt(d[AMX + 1]);
      bool keep = col;
      if (aa_was && fn > kAaSafe * fn_prev) {
        // the accelerated step made the residual worse: clear the ring, back off exponentially
        aa_cnt = 0; aa_head = 0; aa_valid = 0; keep = false;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
        aa_cool = aa_pen;
        aa_pen = aa_pen < 64 ? 2 * aa_pen : 64;
      } else if (aa_cool > 0) --aa_cool;
      if (keep) {
        aa_valid |= 1u << slot;
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < AMX; ++j) {
            if (!((aa_valid >> j) & 1u)) continue;
            AaH[slot * AMX + j] = d[j];
            AaH[j * AMX + slot] = d[j];
            if (j != slot) AaH[AMX * AMX + j] += d[j];   // dF_j . f_k = dF_j . f_(k-1) + dF_j . dF_slot
          }
          AaH[AMX * AMX + slot] = d[AMX];
        }
        aa_head = slot + 1 == aa_m ? 0 : slot + 1;
        aa_cnt = aa_cnt < aa_m ? aa_cnt + 1 : aa_m;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      aa_have_prev = true; fn_prev = fn; aa_was = false;
      real dself = 0;   // |dF_new|^2
#pragma unroll
      for (int j = 0; j < AMX; ++j) dself = j == slot ? d[j] : dself;
      real gam[AMX];
#pragma unroll
      for (int j = 0; j < AMX; ++j) gam[j] = 0;
      // no extrapolation while the map drifts (|dF_new| <= kAaDrift |f|): the differences are rounding noise
      const bool ext = aa_cnt > 0 && aa_cool == 0 && !check && dself > (kAaDrift * kAaDrift) * d[AMX + 1];
      if (ext) {
        // gamma = (H + eta I)^-1 b: Gauss-Jordan on the augmented system spread over the wave, lane 8 i + j holding
        // entry (i, j) (regularised Gram matrix: no pivoting)
        static_assert(AMX <= 7, "one 8 x 8 lane tile holds the augmented system");
        const int gi = lane >> 3, gj = lane & 7;
        real tr = 0;
#pragma unroll
        for (int i = 0; i < AMX; ++i) tr += AaH[i * AMX + i];          // dead slots hold zeros
        const real eta = kAaReg * tr + 1e-300;
        real ae = 0;
        if (gi < AMX && gj <= AMX) ae = gj < AMX ? AaH[gi * AMX + gj] : AaH[AMX * AMX + gi];
        if (gi < AMX && gi == gj) ae = ((aa_valid >> gi) & 1u) ? ae + eta : 1.0;
#pragma unroll
        for (int k = 0; k < AMX; ++k) {
          const real piv = __shfl(ae, 9 * k);
          const real rk = __shfl(ae, 8 * k + gj);
          const real ck = __shfl(ae, 8 * gi + k);
          const real rs = rk / piv;
          ae = gi == k ? rs : ae - ck * rs;
        }
#pragma unroll
        for (int j = 0; j < AMX; ++j) gam[j] = __shfl(ae, 8 * j + AMX);
        aa_was = true;
      }
      // ---- apply: u = g - sum_j gamma_j dG_j; the site rows are projected from their (extrapolated) point -------
      auto aa_apply = [&](real* zdst, unsigned zo, unsigned uo, real (&out)[4]) __attribute__((always_inline)) {
        real gv[4];
        float hv[AMX][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gv[r] = zdst[zo + 64 * r];
        if (ext) {
#pragma unroll
          for (int j = 0; j < AMX; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              hv[j][r] = 0.f;
              if ((aa_valid >> j) & 1u) hv[j][r] = HG[(size_t)j * DU + uo + 64 * r];   // uniform
            }
#pragma unroll
          for (int j = 0; j < AMX; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] -= gam[j] * (real)hv[j][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { UP[uo + 64 * r] = gv[r]; out[r] = gv[r]; }
      };
#pragma unroll 1
      for (int c = wave; c < nct; c += NWV) {
        RELANE();
#pragma unroll 1
        for (int e = 0; e < NE; ++e) {
          real o4[4];
          aa_apply(RZ, fidx(e, c, 0), fidx(e, c, 0), o4);
          if (ext) {
#pragma unroll
            for (int r = 0; r < 4; ++r) RZ[fidx(e, c, r)] = o4[r];
          }
        }
        const int tt = 16 * c + t;
        real pk = M::big;
        if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) {
          real zhr[4], gxn[4], lim[4];
          int ty[4];
          aa_apply(ZHR, sidx2(mo, c, 0), (unsigned)NT + sidx2(mo, c, 0), zhr);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * mo + M::rowof(g, r);
            gxn[r] = GX[sidx2(mo, c, r)]; lim[r] = RLi[j]; ty[r] = RTi[j];
          }
          site_project(mo, c, zhr, gxn, ty, lim, pk);
        }
      }
    }
    __syncthreads();
    STAMP(3);   // barrier 1
    // ================= phase 2: row items =========================================================================
#pragma unroll 1
    for (int ri = wave; ri < 4 * NE; ri += NWV) {
      RELANE();
      const int e = ri >> 2, r = ri & 3;
      real zh[CTL], lbv[CTL], ubv[CTL], z1[CTL];
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        zh[c] = 0; lbv[c] = 0; ubv[c] = 0;
        if (c < nct) {   // uniform
          const unsigned i = fidx(e, c, r);
          zh[c] = RZ[i]; lbv[c] = LBs[i]; ubv[c] = UBs[i];
        }
      }
      STAMP(4);   // row loads issued
      project_row(e, r, zh, lbv, ubv, z1, false);
      STAMP(5);   // water-filling (includes the wait for the loads)
      // x and q of the row as one batch (into the registers of the bounds), then y1, the new r0 and the stores
#pragma unroll
      for (int c = 0; c < CTL; ++c)
        if (c < nct) { const unsigned i = fidx(e, c, r); lbv[c] = Xs[i]; ubv[c] = Qs[i]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        if (c < nct) {
          const unsigned i = fidx(e, c, r);
          const real y1n = rho * (zh[c] - z1[c]);
          Z1s[i] = z1[c]; Y1s[i] = y1n;
          RZ[i] = sigma * lbv[c] - ubv[c] + rho * z1[c] - y1n;   // the new r0
        }
      }
      STAMP(6);   // y1, new r0
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);   // barrier 2
    if (check) {
      // ---- residuals (column items; state re-read: L2-hot); (G' y2) by MFMA with the un-rotated site matrix --------
      real v0 = sv0, v1 = 0, v2 = sv2, v4 = 0, v5 = 0;
#pragma unroll 1
      for (int c = wave; c < nct; c += NWV) {
        RELANE();
#pragma unroll 1
        for (int e = 0; e < NE; ++e) {
          vec4 gty = {0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[sidx2(m, c, s)], gty);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned i = fidx(e, c, r);
            const real xk = Xs[i], qk = Qs[i], yk = Y1s[i], zk = Z1s[i];
            v0 = fmax(v0, fabs(xk - zk));
            v1 = fmax(v1, fabs(pd * xk + qk + yk + gty[r]));
            v2 = fmax(v2, fmax(fabs(xk), fabs(zk)));
            v4 = fmax(v4, fabs(pd * xk));
            v5 = fmax(v5, fabs(yk + gty[r]));
          }
        }
      }
      real v[5] = {v0, v1, v2, v4, v5};
      stream_block_max<5, NWV>(v, SC, lane, wave);
      pri = v[0]; dua = v[1];
      const real npri = v[2], ndua = fmax(fmax(v[3], v[4]), qnorm);
      const real eps_p = A.eps_abs + A.eps_rel * npri, eps_d = A.eps_abs + A.eps_rel * ndua;
      if (pri <= eps_p && dua <= eps_d) { status = 1; done = true; }
      if (!done && have_prev) {
        // ---- primal infeasibility certificate (OSQP's, generalised to the sets B and C; acn_qp_tiled.hpp) ----------
        // v = y - y(previous check).  If A'v ~ 0 and the support function of B x C at v is negative, no point of
        // B x C satisfies A r = z.  For B the support function of a session is bounded above by
        // phi(l) = l cap + sum_t [ub (v_t - l)+ + lb (v_t - l)-] for any admissible l.
        real w6[2] = {0, 0};   // |v|, |v1 + G'v2|
#pragma unroll 1
        for (int c = wave; c < nct; c += NWV) {
          RELANE();
          real dv2[MT][4];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned i = sidx2(m, c, r);
              dv2[m][r] = Y2[i] - (real)Y2P[i];
              w6[0] = fmax(w6[0], fabs(dv2[m][r]));
            }
#pragma unroll 1
          for (int e = 0; e < NE; ++e) {
            vec4 gtv = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int s = 0; s < 4; ++s)
                gtv = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], dv2[m][s], gtv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned i = fidx(e, c, r);
              const real v1 = Y1s[i] - (real)Y1P[i];
              w6[0] = fmax(w6[0], fabs(v1));
              w6[1] = fmax(w6[1], fabs(v1 + gtv[r]));
            }
          }
        }
        stream_block_max<2, NWV>(w6, SC, lane, wave);
        const real vn = w6[0];
        const real vtol = 1e-4 * vn;
        if (vn > 1e-12 * fmax(1.0, qnorm) && w6[1] <= vtol) {   // block-uniform
          real ssum = 0, bad = 0;
          // site rows (column items)
#pragma unroll 1
          for (int c = wave; c < nct; c += NWV) {
            RELANE();
            const int tt = 16 * c + t;
            real pk = M::big;
            if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const unsigned i = sidx2(m, c, r);
                const int j = 16 * m + M::rowof(g, r);
                const real v2 = Y2[i] - (real)Y2P[i];
                const int ty = RTi[j];
                if (ty == kRowBox) { ssum += RLi[j] * fmax(v2, 0.0); if (v2 < -vtol) bad = 1; }
                else if (ty == kRowPeak) {
                  if (pk < M::big) ssum += pk * fmax(v2, 0.0); else if (v2 > vtol) bad = 1;
                  if (v2 < -vtol) bad = 1;
                } else if (ty == kRowSocRe) {
                  const unsigned i2 = sidx2(m, c, (r + 1) & 3);
                  const real vi = Y2[i2] - (real)Y2P[i2];
                  ssum += RLi[j] * sqrt(v2 * v2 + vi * vi);
                } else if (ty == kRowSocIm) {
                } else if (fabs(v2) > vtol) bad = 1;   // free / quadratic rows admit no ray
              }
          }
          // sessions (row items): bound each session's support function; periods outside every window are pinned
          // to lb = ub: support lb * v
#pragma unroll 1
          for (int ri = wave; ri < 4 * NE; ri += NWV) {
            RELANE();
            const int e = ri >> 2, r = ri & 3;
            const int ev = 16 * e + M::rowof(g, r);
            real vv[CTL], lbv[CTL], ubv[CTL];
            bool cov[CTL];
#pragma unroll
            for (int c = 0; c < CTL; ++c) {
              vv[c] = 0; lbv[c] = 0; ubv[c] = 0; cov[c] = false;
              if (c < nct) {
                const unsigned i = fidx(e, c, r);
                vv[c] = Y1s[i] - (real)Y1P[i]; lbv[c] = LBs[i]; ubv[c] = UBs[i];
              }
            }
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
              const size_t sidx = ((size_t)b * K + k) * N + (ev < N ? ev : 0);
              const int off = ev < N ? A.s_off[sidx] : 0;
              int len = ev < N ? A.s_len[sidx] : 0;
              if (off + len > Tm) len = Tm - off;
              const real cap = ev < N ? A.s_cap[sidx] : 0.0;
              real lmin_l = M::big, lmax_l = -M::big;
#pragma unroll
              for (int c = 0; c < CTL; ++c) {
                const int tp = 16 * c + t;
                const bool inw = tp >= off && tp < off + len;
                cov[c] = cov[c] || inw;
                lmin_l = inw ? fmin(lmin_l, vv[c]) : lmin_l;
                lmax_l = inw ? fmax(lmax_l, vv[c]) : lmax_l;
              }
              real lam3[3];
              lam3[0] = row_min<real>(lmin_l);
              lam3[1] = row_max<real>(lmax_l);
              lam3[2] = 0;
              real best = M::big;
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                real l_ = lam3[j];
                if (!eq) l_ = fmax(l_, 0.0);
                real ph = 0;
#pragma unroll
                for (int c = 0; c < CTL; ++c) {
                  const int tp = 16 * c + t;
                  if (tp >= off && tp < off + len) {
                    const real dv = vv[c] - l_;
                    ph += ubv[c] * fmax(dv, 0.0) + lbv[c] * fmin(dv, 0.0);
                  }
                }
                ph = row_sum<real>(ph) + l_ * cap;
                best = fmin(best, ph);
              }
              if (len > 0 && t == 0) ssum += best;   // one lane per session
            }
#pragma unroll
            for (int c = 0; c < CTL; ++c)
              if (!cov[c]) ssum += lbv[c] * vv[c];
          }
          const real tot = wave_sum<real>(ssum), bw = wave_max<real>(bad);
          if (lane == 0) { SC[wave * 8] = tot; SC[wave * 8 + 1] = bw; }
          __syncthreads();
          real stot = 0, bmax = 0;
          for (int wv = 0; wv < NWV; ++wv) { stot += SC[wv * 8]; bmax = fmax(bmax, SC[wv * 8 + 1]); }
          __syncthreads();
          if (bmax == 0.0 && stot < -vtol) { status = 3; done = true; }
        }
      }
      if (!done) {   // snapshot for the next certificate test
#pragma unroll 1
        for (int c = wave; c < nct; c += NWV) {
          RELANE();
#pragma unroll 1
          for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned i = fidx(e, c, r); Y1P[i] = (float)Y1s[i]; }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(m, c, r); Y2P[i] = (float)Y2[i]; }
        }
        have_prev = true;
      }
      if (done) {
      } else if (it >= A.max_iter) {
        done = true;
        if (pri <= kInaccurate * eps_p && dua <= kInaccurate * eps_d) status = 5;
      } else if (A.adapt_every > 0 && it % A.adapt_every == 0) {
        const real sp = pri / fmax(npri, 1e-12), sd = dua / fmax(ndua, 1e-12);
        const real ratio = sqrt(sp / fmax(sd, 1e-30));
        const real tol_eff = A.adapt_tol * (1.0 + (real)n_adapt * (1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = fmin(fmax(rho * ratio, 1e-6), 1e6);
#pragma unroll 1
          for (int c = wave; c < nct; c += NWV) { RELANE(); rebuild_r0(c); if (aa_m > 0) reset_u(c); }   // r0 and u depend on rho; own columns
          if (aa_m > 0) {   // the fixed-point map changed: restart the ring from the current (z, y)
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
          }
        }
      }
    }
    STAMP(8);   // residual check (amortised)
  }
#ifdef ACNQP_STAMPS
  if (lane == 0 && b < 1024 && wave < 16)
    for (int k = 0; k < 12; ++k) g_stamps[(b * 16 + wave) * 12 + k] = st_acc[k];
#endif

  // ---- results: the feasible iterate z1 is the schedule --------------------------------------------------------
  real ol = 0;
#pragma unroll 1
  for (int c = wave; c < nct; c += NWV)
#pragma unroll 1
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
        if (ev < N && tt < Tm) {
          const unsigned i = fidx(e, c, r);
          const real z = Z1s[i];
          A.x[((size_t)b * N + ev) * Tm + tt] = z;
          ol += (0.5 * pd_user * z + Qs[i]) * z;
        }
      }
  if (A.y_out) {   // site-row multipliers in the caller's row order and units
#pragma unroll 1
    for (int c = wave; c < nct; c += NWV)
#pragma unroll
      for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
          const int ja = A.rowabi[j];
          if (ja >= 0 && tt < Tm)
            A.y_out[((size_t)b * A.Mg + ja) * Tm + tt] = Y2[sidx2(mo, c, r)] * static_cast<const real*>(A.rowscale)[j];
        }
  }
  ol = wave_sum<real>(ol);
  if (lane == 0) SC[wave] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < NWV; ++wv) o += SC[wv];
    A.status[b] = status; A.iters[b] = it; A.pri[b] = pri; A.dua[b] = dua; A.obj[b] = o;
  }
}

#undef RELANE
}  // namespace acnqp
