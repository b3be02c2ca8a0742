// Long-horizon kernel of the batched MPC QP solver for gfx950: horizons of 33 ... 288 periods (the reference's
// N = 54 x T = 144 stress scenarios, tests/test_adacharge_stress.py; day-long offline problems, aco.py:403-408),
// any site of up to 32 padded rows -- the shapes neither the register-resident kernel (T <= 32) nor the large-site
// kernel (T <= 48, whole rows of every tile in registers AND LDS slabs sized by the horizon) can hold.
//
// Same ADMM as acn_qp_tiled.hpp / acn_qp_stream.hpp.  What changes is the blocking.  Only ONE step of the iteration
// couples the periods of an EVSE: the projection onto its energy rows (water-filling over the session window).
// Everything else -- P = Ghat r0, the eigen-space step, the site-row projection, x~ = (r0 + Ghat' e^) / a -- is
// independent per period, so an iteration runs as
//
//   phase 1, per COLUMN BLOCK of 48 periods (kLongCB = 3 column tiles):
//     (a) every wave: partial P of its own EVSE tiles for the block (MFMA, r0 read back from the workspace) -> LDS
//         barrier
//     (b) the block's MT x 3 site tiles, dealt round-robin to the waves: sum of the partials, e^, h^ -> LDS
//         barrier
//     (c) same tiles: G x~ = Q h^, relaxation, projection onto C, y2                      (no barrier: (d) needs e^ only)
//     (d) every wave, own EVSE tiles: x~ (MFMA with e^), relaxation, zh -> workspace (over r0, which (a) consumed)
//   phase 2, per EVSE tile and register row (= 4 EVSEs x the WHOLE horizon, CTL column registers per lane):
//     zh, lb, ub -> water-filling (safeguarded Newton along the 16-lane DPP rows) -> z1, y1, the new r0 -> workspace
//
// LDS holds one column block (24 MT + 12 MT KB), registers hold one row (4 x CTL doubles): both independent of N, and
// the horizon only sets the number of column blocks and the row length.  State streams through a per-problem
// workspace in MFMA fragment order (L2 / MALL resident at these sizes: 54 x 144 is 74 KB per array).
// Two barriers per column block; the cross-wave data (partials, e^, h^) never leave LDS.
//
// Not in this kernel (the general-shape kernel keeps them): the demand-charge row (its prox couples the periods of a
// SITE row), the infeasibility certificate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acn_qp_stream.hpp"

namespace acnqp {

constexpr int kLongCB = 3;      // column tiles per block (48 periods)
constexpr int kLongWaves = 4;

// doubles of workspace one problem needs
__host__ __device__ inline long long long_workspace(int NP, int CTL, int K, int MT) {
  const long long NT = (long long)(NP / 16) * CTL * 256;
  return 7 * NT + (long long)K * NP + 3LL * MT * CTL * 256 + 64;
}

// LDS carve-up (doubles)
struct LongLds {
  int part, g0h, we, scal, total;
  __host__ __device__ explicit LongLds(int MT) {
    int o = 0;
    part = o; o += kLongWaves * MT * kLongCB * 256;   // per-wave partial P of the block
    g0h = o;  o += MT * kLongCB * 256;                // Ghat z1 (start) / h^
    we = o;   o += MT * kLongCB * 256;                // e^
    scal = o; o += kLongWaves * 8 + 8;
    total = o;
  }
};

template <int CTL, int MT>
__global__ __launch_bounds__(kLongWaves * 64, CTL <= 9 ? 2 : 1) void admm_long_kernel(const StreamArgs SA) {
  constexpr int CB = kLongCB;
  constexpr int NWV = kLongWaves;
  static_assert(CTL % CB == 0, "whole column blocks");
  using M = Mfma<double>;
  using vec4 = M::vec4;
  typedef double real;
  const TiledArgs& A = SA.t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* sm = reinterpret_cast<real*>(smem_raw);
  const LongLds L(MT);
  real* PART = sm + L.part;
  real* G0H = sm + L.g0h;
  real* WE = sm + L.we;
  real* SC = sm + L.scal;

  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
  int g = lane >> 4, t = lane & 15;
#define RELANE() do { asm volatile("" : "+v"(lane)); g = lane >> 4; t = lane & 15; } while (0)
  const int N = A.N, Tm = A.Tm, NP = A.NP, K = A.K;
  const int NE = NP >> 4;                 // EVSE tiles
  const int nct = (Tm + 15) >> 4;         // column tiles that hold periods (<= CTL); the rest is padding, all zero
  const int ncb = (nct + CB - 1) / CB;    // column blocks phase 1 visits
  const long long NT = (long long)NE * CTL * 256;
  real* W0 = SA.work + (size_t)b * SA.ws_per_problem;
  real *Xs = W0, *Z1s = Xs + NT, *Y1s = Z1s + NT, *Qs = Y1s + NT, *LBs = Qs + NT, *UBs = LBs + NT;
  real* RZ = UBs + NT;                    // r0 (read by phase 1 (a), (d)) / zh (written by (d), read by phase 2)
  real* MU = RZ + NT;                     // [K][NP]
  real* Z2 = MU + (size_t)K * NP;         // site-row state in tile-fragment order [MT][CTL][4][64]
  real* Y2 = Z2 + MT * CTL * 256;
  real* GX = Y2 + MT * CTL * 256;
  const real* FG = static_cast<const real*>(A.fragG);
  const real* FQ = static_cast<const real*>(A.fragQ);
  const real* Gm = static_cast<const real*>(A.G);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);
  const bool eq = A.s_eq[b] != 0;
  const real sigma = A.sigma, alpha = A.alpha;
  const real lfb = A.lf ? A.lf[b] / (A.flat_scale * A.flat_scale) : 0.0;

  auto fidx = [&](int e, int c, int r) -> size_t { return ((size_t)(e * CTL + c) * 4 + r) * 64 + lane; };

  // ---- init: inputs -> fragment order, one register row at a time; |q|_inf, max ub; a session whose bounds cannot
  // meet its energy row -----------------------------------------------------------------------------------------
  real qn = 0, um = 0, bad = 0;
#pragma unroll 1
  for (int e = wave; e < NE; e += NWV) {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      RELANE();
      const int ev = 16 * e + M::rowof(g, r);
      real lbv[CTL], ubv[CTL];
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const int tt = 16 * c + t;
        const bool ok = ev < N && tt < Tm;
        const size_t idx = ((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0);
        const real l = ok ? A.lb[idx] : 0.0;
        real u = ok ? A.ub[idx] : 0.0;
        const real q = ok ? A.q[idx] : 0.0;
        if (u < l) u = l;
        lbv[c] = l; ubv[c] = u;
        const size_t i = fidx(e, c, r);
        LBs[i] = l; UBs[i] = u; Qs[i] = q; RZ[i] = 0;
        qn = fmax(qn, fabs(q)); um = fmax(um, u);
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

  // (a): this wave's partial Ghat[:, own tiles] v[own tiles] for column block cb, v = the RZ array (r0, or z1 during
  // the start), into PART[wave]
  auto partial_p = [&](int cb) __attribute__((always_inline)) {
    vec4 acc[MT][CB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) acc[m][cc] = vec4{0, 0, 0, 0};
#pragma unroll 1
    for (int e = wave; e < NE; e += NWV) {
      RELANE();
      const real* fg = FG + (size_t)e * MT * 2 * 4 * 64;
      real af[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) af[m][s] = fg[((m * 2 + 0) * 4 + s) * 64 + lane];
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        real bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[s] = RZ[fidx(e, cb * CB + cc, s)];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[m][cc] = M::mma(af[m][s], bv[s], acc[m][cc]);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int cc = 0; cc < CB; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) PART[(((wave * MT + m) * CB + cc) * 4 + r) * 64 + lane] = acc[m][cc][r];
  };
  // sum of the waves' partials for site tile (mo, cc) of the block, in wave order
  auto sum_part = [&](int mo, int cc, real (&g0)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      real s = PART[(((0 * MT + mo) * CB + cc) * 4 + r) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NWV; ++w) s += PART[(((w * MT + mo) * CB + cc) * 4 + r) * 64 + lane];
      g0[r] = s;
    }
  };
  // r0 of this wave's tiles from the stored state (start, and after a rho change)
  auto rebuild_r0 = [&]() __attribute__((always_inline)) {
#pragma unroll 1
    for (int e = wave; e < NE; e += NWV) {
      RELANE();
#pragma unroll 1
      for (int c = 0; c < nct; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t i = fidx(e, c, r);
          RZ[i] = sigma * Xs[i] - Qs[i] + rho * Z1s[i] - Y1s[i];
        }
    }
  };

  // ---- start (see acn_qp_tiled.hpp): z1 = Proj_B(-kStartGain q), x = z1, y1 = -(q + pd z1); z2 = G z1 = Q (Ghat z1);
  // warm (optional): z1 = Proj_B(warm_x), y2 = warm_y, y1 = -(q + pd z1 + G' y2) -------------------------------------
  const bool warm = A.warm_x != nullptr && A.warm_y != nullptr;
#pragma unroll 1
  for (int e = wave; e < NE; e += NWV) {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      RELANE();
      real zs[CTL], lbv[CTL], ubv[CTL], z1[CTL];
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const size_t i = fidx(e, c, r);
        lbv[c] = LBs[i]; ubv[c] = UBs[i];
        zs[c] = -kStartGain * Qs[i];
        if (warm) {
          const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
          const bool ok = ev < N && tt < Tm;
          zs[c] = ok ? A.warm_x[((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0)] : 0.0;
        }
      }
      project_row(e, r, zs, lbv, ubv, z1, true);
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const size_t i = fidx(e, c, r);
        Xs[i] = z1[c]; Z1s[i] = z1[c]; Y1s[i] = -(Qs[i] + pd * z1[c]); RZ[i] = z1[c];
      }
    }
  }
#pragma unroll 1
  for (int cb = 0; cb < ncb; ++cb) {
    partial_p(cb);
    __syncthreads();
#pragma unroll 1
    for (int tl = wave; tl < MT * CB; tl += NWV) {
      RELANE();
      const int mo = tl / CB, cc = tl - mo * CB;
      real g0[4];
      sum_part(mo, cc, g0);
#pragma unroll
      for (int r = 0; r < 4; ++r) G0H[((mo * CB + cc) * 4 + r) * 64 + lane] = g0[r];
    }
    __syncthreads();
#pragma unroll 1
    for (int tl = wave; tl < MT * CB; tl += NWV) {
      RELANE();
      const int mo = tl / CB, cc = tl - mo * CB, c = cb * CB + cc;
      vec4 zt = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          zt = M::mma(FQ[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], G0H[((mi * CB + cc) * 4 + s) * 64 + lane], zt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ((mo * CTL + c) * 4 + r) * 64 + lane;
        real yv = 0;
        if (warm) {
          const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
          const int ja = A.rowabi[j];
          if (ja >= 0 && tt < Tm) yv = A.warm_y[((size_t)b * A.Mg + ja) * Tm + tt] / static_cast<const real*>(A.rowscale)[j];
        }
        Z2[i] = zt[r]; GX[i] = zt[r]; Y2[i] = yv;
      }
    }
    // the next block's partials go to PART (last read before the barrier above); its G0H writes come after the
    // barrier that follows them: no barrier needed here
  }
  __syncthreads();
  if (warm) {   // y1 = -(q + pd z1 + G' y2): the cold start stored the G' y2 = 0 version
#pragma unroll 1
    for (int e = wave; e < NE; e += NWV) {
      RELANE();
#pragma unroll 1
      for (int c = 0; c < nct; ++c) {
        vec4 gty = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[((m * CTL + c) * 4 + s) * 64 + lane], gty);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y1s[fidx(e, c, r)] -= gty[r];
      }
    }
  }
  rebuild_r0();

  int status = 2, it = 0, n_adapt = 0;
  real pri = M::big, dua = M::big;
  bool done = false;
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
    const real* Lmi = Lm + zoff;
    const real* RLi = RL + zoff;
    const int32_t* RTi = A.rowtype + zoff;
    real sv0 = 0, sv2 = 0;   // |G x - z2|_inf, max(|G x|, |z2|): the site-row share of the residuals
    // ================= phase 1: column blocks =================================================================
#pragma unroll 1
    for (int cb = 0; cb < ncb; ++cb) {
      partial_p(cb);
      __syncthreads();
      // ---- (b) eigen space: e^ -> WE, h^ -> G0H -------------------------------------------------------------
#pragma unroll 1
      for (int tl = wave; tl < MT * CB; tl += NWV) {
        RELANE();
        const int mo = tl / CB, cc = tl - mo * CB, c = cb * CB + cc;
        real g0[4];
        sum_part(mo, cc, g0);
        vec4 wh = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int i = ((mi * CTL + c) * 4 + s) * 64 + lane;
            wh = M::mma(FQi[(((mo * MT + mi) * 2 + 0) * 4 + s) * 64 + lane], rho * Z2[i] - Y2[i], wh);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real lj = Lmi[16 * mo + M::rowof(g, r)];
          const real e_ = wh[r] - (rho / (a + rho * lj)) * (g0[r] + lj * wh[r]);
          WE[((mo * CB + cc) * 4 + r) * 64 + lane] = e_;
          G0H[((mo * CB + cc) * 4 + r) * 64 + lane] = (g0[r] + lj * e_) * inv_a;
        }
      }
      __syncthreads();
      // ---- (c) site rows: G x~ = Q h^, relaxation, projection onto C, y2 -------------------------------------
#pragma unroll 1
      for (int tl = wave; tl < MT * CB; tl += NWV) {
        RELANE();
        const int mo = tl / CB, cc = tl - mo * CB, c = cb * CB + cc;
        vec4 zt = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            zt = M::mma(FQi[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], G0H[((mi * CB + cc) * 4 + s) * 64 + lane], zt);
        real zhr[4], lim[4];
        int ty[4];
        const int tt = 16 * c + t;
        real pk = M::big;
        if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ((mo * CTL + c) * 4 + r) * 64 + lane;
          const int j = 16 * mo + M::rowof(g, r);
          ty[r] = RTi[j]; lim[r] = RLi[j];
          GX[i] = alpha * zt[r] + (1.0 - alpha) * GX[i];
          zhr[r] = alpha * zt[r] + (1.0 - alpha) * Z2[i] + Y2[i] * inv_rho;
        }
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
          const int i = ((mo * CTL + c) * 4 + r) * 64 + lane;
          real zn = zhr[r];
          if (ty[r] == kRowBox) zn = fmin(zn, lim[r]);
          else if (ty[r] == kRowPeak) zn = fmin(zn, pk);
          else if (ty[r] == kRowQuad) zn = zn * (rho / (rho + lfb));
          else if (ty[r] == kRowSocRe || ty[r] == kRowSocIm) zn = zn * scl[r >> 1];
          Y2[i] = rho * (zhr[r] - zn);
          Z2[i] = zn;
          sv0 = fmax(sv0, fabs(GX[i] - zn));
          sv2 = fmax(sv2, fmax(fabs(GX[i]), fabs(zn)));
        }
      }
      // ---- (d) x~ of this wave's tiles for the block; zh takes r0's place ---------------------------------------
#pragma unroll 1
      for (int e = wave; e < NE; e += NWV) {
        RELANE();
        const real* fg = FG + (size_t)e * MT * 2 * 4 * 64;
        real fx[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) fx[m][s] = fg[((m * 2 + 1) * 4 + s) * 64 + lane];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
          const int c = cb * CB + cc;
          real xv[4], z1o[4], y1o[4];
          vec4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t i = fidx(e, c, r);
            acc[r] = RZ[i]; xv[r] = Xs[i]; z1o[r] = Z1s[i]; y1o[r] = Y1s[i];
          }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc = M::mma(fx[m][s], WE[((m * CB + cc) * 4 + s) * 64 + lane], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t i = fidx(e, c, r);
            const real xn = acc[r] * inv_a;
            RZ[i] = alpha * xn + (1.0 - alpha) * z1o[r] + y1o[r] * inv_rho;
            Xs[i] = alpha * xn + (1.0 - alpha) * xv[r];
          }
        }
      }
      // (a) of the next block writes PART (last read in (b), two barriers back); its (b) writes WE / G0H after the
      // barrier that follows (a): nothing to wait for here
    }
    __syncthreads();   // every site tile's y2 is stored before the residual check reads it across waves
    // ================= phase 2: whole-horizon rows of this wave's tiles ========================================
    real v0 = sv0, v1 = 0, v2 = sv2, v4 = 0, v5 = 0;
#pragma unroll 1
    for (int e = wave; e < NE; e += NWV) {
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {
        RELANE();
        real zh[CTL], lbv[CTL], ubv[CTL], z1[CTL];
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          zh[c] = 0; lbv[c] = 0; ubv[c] = 0;
          if (c < nct) {   // uniform: padding tiles hold zeros and stay zero
            const size_t i = fidx(e, c, r);
            zh[c] = RZ[i]; lbv[c] = LBs[i]; ubv[c] = UBs[i];
          }
        }
        project_row(e, r, zh, lbv, ubv, z1, false);
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          if (c < nct) {
            const size_t i = fidx(e, c, r);
            const real y1n = rho * (zh[c] - z1[c]);
            Z1s[i] = z1[c]; Y1s[i] = y1n;
            RZ[i] = sigma * Xs[i] - Qs[i] + rho * z1[c] - y1n;   // the new r0
          }
        }
      }
      if (check) {   // residual terms of this tile (state re-read: L2-hot); (G' y2) tile by MFMA with the un-rotated site matrix
        RELANE();
#pragma unroll 1
        for (int c = 0; c < nct; ++c) {
          vec4 gty = {0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[((m * CTL + c) * 4 + s) * 64 + lane], gty);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t i = fidx(e, c, r);
            const real xk = Xs[i], qk = Qs[i], yk = Y1s[i], zk = Z1s[i];
            v0 = fmax(v0, fabs(xk - zk));
            v1 = fmax(v1, fabs(pd * xk + qk + yk + gty[r]));
            v2 = fmax(v2, fmax(fabs(xk), fabs(zk)));
            v4 = fmax(v4, fabs(pd * xk));
            v5 = fmax(v5, fabs(yk + gty[r]));
          }
        }
      }
    }
    if (check) {
      real v[5] = {v0, v1, v2, v4, v5};
      stream_block_max<5, NWV>(v, SC, lane, wave);
      pri = v[0]; dua = v[1];
      const real npri = v[2], ndua = fmax(fmax(v[3], v[4]), qnorm);
      const real eps_p = A.eps_abs + A.eps_rel * npri, eps_d = A.eps_abs + A.eps_rel * ndua;
      if (pri <= eps_p && dua <= eps_d) { status = 1; done = true; }
      else if (it >= A.max_iter) {
        done = true;
        if (pri <= kInaccurate * eps_p && dua <= kInaccurate * eps_d) status = 5;
      } else if (A.adapt_every > 0 && it % A.adapt_every == 0) {
        const real sp = pri / fmax(npri, 1e-12), sd = dua / fmax(ndua, 1e-12);
        const real ratio = sqrt(sp / fmax(sd, 1e-30));
        const real tol_eff = A.adapt_tol * (1.0 + (real)n_adapt * (1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = fmin(fmax(rho * ratio, 1e-6), 1e6);
          rebuild_r0();        // r0 depends on rho; own tiles only, no barrier needed
        }
      }
    }
  }

  // ---- results: the feasible iterate z1 is the schedule --------------------------------------------------------
  __syncthreads();
  real ol = 0;
#pragma unroll 1
  for (int e = wave; e < NE; e += NWV)
#pragma unroll 1
    for (int c = 0; c < nct; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
        if (ev < N && tt < Tm) {
          const size_t i = fidx(e, c, r);
          const real z = Z1s[i];
          A.x[((size_t)b * N + ev) * Tm + tt] = z;
          ol += (0.5 * pd_user * z + Qs[i]) * z;
        }
      }
  if (A.y_out) {   // site-row multipliers in the caller's row order and units
#pragma unroll 1
    for (int tl = wave; tl < MT * nct; tl += NWV) {
      const int mo = tl / nct, c = tl - mo * nct;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
        const int ja = A.rowabi[j];
        if (ja >= 0 && tt < Tm)
          A.y_out[((size_t)b * A.Mg + ja) * Tm + tt] = Y2[((mo * CTL + c) * 4 + r) * 64 + lane] * static_cast<const real*>(A.rowscale)[j];
      }
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
