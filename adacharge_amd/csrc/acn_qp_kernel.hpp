// Batched MPC QP solver for gfx950: one 256-thread workgroup per problem.
//
// Device algorithm: OSQP-form ADMM with the per-session energy rows folded
// into the projection set (restated in numpy in oracle/admm_ref.py):
//
//   x~   = (a I + rho G'G)^-1 (sigma x - q + rho z1 - y1 + G'(rho z2 - y2))   per period
//   z1   = Proj_B (alpha x~ + (1-alpha) z1 + y1/rho),   B = box /\ session energy rows
//   z2   = Proj_C (alpha G x~ + (1-alpha) z2 + y2/rho), C = site rows (box / disc / peak)
//   y    = rho (pre-projection point - projected point)
//
// Thread roles inside a workgroup (N <= 64 EVSEs, Tm <= 4*TPT periods):
//   variable role   tid = 4*i + tq : EVSE i, periods [tq*TPT, (tq+1)*TPT).  x, z1, y1, q,
//                   lb, ub live in registers; a session's energy sum is a quad reduction
//                   (2 cross-lane adds), so the water-filling projection needs no LDS.
//   eigen role      (j, t) of the rotated site rows Ghat = Q'G: the only cross-EVSE
//                   reduction of an iteration, Ghat r0[:, t], read from an LDS copy of r0.
//   constraint role (c, t) of the site rows / SOC pairs: z2, y2, G x in LDS, one owner each.
// Two workgroup barriers per iteration; the site matrices live in LDS for the whole solve,
// so HBM is touched once per problem (load lb/ub/q/sessions, store the schedule).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acnqp {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxEvse = kThreads / 4;
constexpr int kMaxRows = 40;   // rows of G
constexpr int kMaxK = 4;       // session slots per EVSE
constexpr int kNumRed = 8;

struct KernelArgs {
  int B, N, Tm, K, Mg, M, Mc, cone, has_peak;
  const double *G, *Ghat, *Q, *lam, *limits;   // site, device
  const int32_t* horizon;
  const double *lb, *ub, *q, *pdiag;
  const int32_t *s_off, *s_len;
  const double* s_cap;
  const uint8_t* s_eq;
  const double* peak;
  double* x;
  int32_t *status, *iters;
  double *pri, *dua, *obj;
  double eps_abs, eps_rel, rho0, sigma, alpha, adapt_tol, reg_rel;
  int max_iter, check_every, adapt_every;
};

__host__ __device__ inline int odd_stride(int n) { return n | 1; }

// LDS carve-up (in units of `real`), shared by host (size) and device (offsets).
struct LdsLayout {
  int mgp, rs;
  int ght, gt, qa, qb, lam, lim, r0, w, e, hh, y2, z2, gx, pk, red, total;
  __host__ __device__ LdsLayout(int N, int Tm4, int Mg, int M) {
    mgp = odd_stride(Mg);
    rs = odd_stride(N > 64 ? N : 64);
    int o = 0;
    ght = o; o += N * mgp;
    gt = o;  o += N * mgp;
    qa = o;  o += Mg * mgp;
    qb = o;  o += Mg * mgp;
    lam = o; o += Mg + 1;
    lim = o; o += M + 1;
    r0 = o;  o += Tm4 * rs;
    w = o;   o += Tm4 * mgp;
    e = o;   o += Tm4 * mgp;
    hh = o;  o += Tm4 * mgp;
    y2 = o;  o += Tm4 * mgp;
    z2 = o;  o += Tm4 * mgp;
    gx = o;  o += Tm4 * mgp;
    pk = o;  o += Tm4 + 1;
    red = o; o += kWaves * kNumRed + 8;
    total = (o + 1) & ~1;
  }
};

template <typename real> __device__ inline real shfl_xor_r(real v, int m) { return __shfl_xor(v, m); }

template <typename real> __device__ inline real quad_sum(real v) {
  v += shfl_xor_r(v, 1);
  v += shfl_xor_r(v, 2);
  return v;
}
template <typename real> __device__ inline real quad_min(real v) {
  v = fmin(v, shfl_xor_r(v, 1));
  v = fmin(v, shfl_xor_r(v, 2));
  return v;
}
template <typename real> __device__ inline real quad_max(real v) {
  v = fmax(v, shfl_xor_r(v, 1));
  v = fmax(v, shfl_xor_r(v, 2));
  return v;
}
template <typename real> __device__ inline real wave_max(real v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_r(v, m));
  return v;
}
template <typename real> __device__ inline real wave_sum(real v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_r(v, m);
  return v;
}

template <typename real> struct Tol;
template <> struct Tol<double> { static constexpr double proj = 1e-13; static constexpr double big = 1e300; };
template <> struct Tol<float>  { static constexpr float proj = 2e-6f;  static constexpr float big = 1e30f; };

// Block-wide max of up to kNumRed per-thread values; every thread gets the result.
// Two barriers.  `Red` is [kWaves][kNumRed].
template <typename real, int NV>
__device__ inline void block_max(real (&v)[NV], real* Red, int lane, int wave) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const real m = wave_max<real>(v[k]);
    if (lane == 0) Red[wave * kNumRed + k] = m;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    real m = Red[k];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) m = fmax(m, Red[wv * kNumRed + k]);
    v[k] = m;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
template <typename real, int TPT, int KS>
__global__ __launch_bounds__(kThreads) void admm_kernel(const KernelArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* sm = reinterpret_cast<real*>(smem_raw);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int N = A.N, Tm = A.Tm, Mg = A.Mg, M = A.M, Mc = A.Mc;
  constexpr int Tm4 = 4 * TPT;
  const LdsLayout L(N, Tm4, Mg, M);
  const int MgP = L.mgp, RS = L.rs;
  real* Ght = sm + L.ght;   // [i][MgP]   Ghat transposed
  real* Gt = sm + L.gt;     // [i][MgP]   G transposed
  real* Qa = sm + L.qa;     // [r][k] = Q[r][k]
  real* Qb = sm + L.qb;     // [k][r] = Q[r][k]
  real* Lam = sm + L.lam;
  real* Lim = sm + L.lim;
  real* R0 = sm + L.r0;     // [t][RS]
  real* W = sm + L.w;       // [t][MgP]   rho z2 - y2
  real* E = sm + L.e;       // [t][MgP]   e^ (rotated)
  real* Hh = sm + L.hh;     // [t][MgP]   Ghat x~ (rotated)
  real* Y2 = sm + L.y2;     // [t][MgP]   y2 (constraint-role state; read by the dual residual)
  real* Z2 = sm + L.z2;     // [t][MgP]   z2 (constraint-role state)
  real* GX = sm + L.gx;     // [t][MgP]   G x (constraint-role state)
  real* Pk = sm + L.pk;     // [t]        peak limit of this problem
  real* Red = sm + L.red;

  // ---- site -> LDS -------------------------------------------------------------------
  for (int idx = tid; idx < N * Mg; idx += kThreads) {
    const int i = idx / Mg, j = idx - i * Mg;
    Ght[i * MgP + j] = (real)A.Ghat[j * N + i];
    Gt[i * MgP + j] = (real)A.G[j * N + i];
  }
  for (int idx = tid; idx < Mg * Mg; idx += kThreads) {
    const int r = idx / Mg, k = idx - r * Mg;
    const real v = (real)A.Q[idx];
    Qa[r * MgP + k] = v;
    Qb[k * MgP + r] = v;
  }
  for (int idx = tid; idx < Mg; idx += kThreads) Lam[idx] = (real)A.lam[idx];
  for (int idx = tid; idx < M; idx += kThreads) Lim[idx] = (real)A.limits[idx];
  for (int idx = tid; idx < Tm4 * MgP; idx += kThreads) { W[idx] = 0; E[idx] = 0; Hh[idx] = 0; Y2[idx] = 0; Z2[idx] = 0; GX[idx] = 0; }
  for (int idx = tid; idx < Tm4; idx += kThreads) {
    double pv = 1e300;
    if (A.has_peak && A.peak && idx < Tm) pv = A.peak[(size_t)b * Tm + idx];
    Pk[idx] = pv < (double)Tol<real>::big ? (real)pv : Tol<real>::big;
  }
  for (int idx = tid; idx < Tm4 * RS; idx += kThreads) R0[idx] = 0;

  // ---- variable role: problem data -> registers ---------------------------------------
  const int ev = tid >> 2, tq = tid & 3;
  const bool vact = ev < N;
  const int t0 = tq * TPT;
  real x[TPT], z1[TPT], y1[TPT], qv[TPT], lbv[TPT], ubv[TPT], r0v[TPT];
  {
    const size_t base = ((size_t)b * N + (vact ? ev : 0)) * Tm;
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) {
      const int t = t0 + tt;
      const bool ok = vact && t < Tm;
      lbv[tt] = ok ? (real)A.lb[base + t] : (real)0;
      ubv[tt] = ok ? (real)A.ub[base + t] : (real)0;
      qv[tt] = ok ? (real)A.q[base + t] : (real)0;
      if (ubv[tt] < lbv[tt]) ubv[tt] = lbv[tt];
      x[tt] = 0; z1[tt] = 0; y1[tt] = 0;
    }
  }
  const bool eq = A.s_eq[b] != 0;
  unsigned wmask[KS];
  bool shas[KS];
  real scap[KS], slo[KS], shi[KS], mu[KS];
  bool empty_set = false;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    wmask[k] = 0; shas[k] = false; scap[k] = 0; slo[k] = 0; shi[k] = 0; mu[k] = 0;
    if (k < A.K) {   // block-uniform
      const size_t sidx = ((size_t)b * A.K + k) * N + (vact ? ev : 0);
      const int off = vact ? A.s_off[sidx] : 0;
      const int len = vact ? A.s_len[sidx] : 0;
      real sl = 0, su = 0;
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) {
        const int t = t0 + tt;
        if (t >= off && t < off + len && t < Tm) { wmask[k] |= 1u << tt; sl += lbv[tt]; su += ubv[tt]; }
      }
      slo[k] = quad_sum<real>(sl);
      shi[k] = quad_sum<real>(su);
      scap[k] = vact ? (real)A.s_cap[sidx] : (real)0;
      shas[k] = len > 0;   // quad-uniform: the four lanes of a quad read the same (slot, EVSE)
      if (shas[k]) {
        const real slack = (real)64 * Tol<real>::proj * fmax((real)1, fabs(scap[k]));
        if (slo[k] > scap[k] + slack) empty_set = true;
        if (eq && shi[k] < scap[k] - slack) empty_set = true;
      }
    }
  }

  // ---- constraint role / eigen role bookkeeping ----------------------------------------
  const int Rc = Mc * Tm;            // constraint roles (c, t), c fastest
  const int Re = Mg * Tm;            // eigen roles (j, t), j fastest
  int H = 1;
  if (Re * 4 <= kThreads) H = 4; else if (Re * 2 <= kThreads) H = 2;
  const int erole_per_pass = kThreads / H;
  const int hsub = tid % H;
  const bool soc = A.cone == 1;
  const real pd_user = (real)A.pdiag[b];
  const real sigma = (real)A.sigma, alpha = (real)A.alpha;
  real rho = (real)A.rho0;

  __syncthreads();
  real qnorm, pd;
  {
    real f[3];
    f[0] = empty_set ? (real)1 : (real)0;
    f[1] = 0;
    f[2] = 0;
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) { f[1] = fmax(f[1], fabs(qv[tt])); f[2] = fmax(f[2], ubv[tt]); }
    block_max<real, 3>(f, Red, lane, wave);
    qnorm = f[1];
    // scale-free Tikhonov floor (exact regularisation of LP instances)
    pd = pd_user;
    if (f[2] > 0) pd = fmax(pd_user, (real)A.reg_rel * qnorm / f[2]);
    if (f[0] > 0) {
      // a session cannot meet its energy row inside its own bounds: nothing to iterate on
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) {
        const int t = t0 + tt;
        if (vact && t < Tm) A.x[((size_t)b * N + ev) * Tm + t] = 0;
      }
      if (tid == 0) {
        A.status[b] = 4; A.iters[b] = 0;
        A.pri[b] = (double)Tol<real>::big; A.dua[b] = (double)Tol<real>::big; A.obj[b] = 0;
      }
      return;
    }
  }

  // r0 for the first iteration: sigma x - q + rho z1 - y1 with x = z1 = y1 = 0
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    r0v[tt] = -qv[tt];
    if (vact) R0[(t0 + tt) * RS + ev] = r0v[tt];
  }
  __syncthreads();

  int status = 2, it = 0;
  real pri = Tol<real>::big, dua = Tol<real>::big;
  bool done = false;

  while (!done) {
    ++it;
    const real a = sigma + pd + rho;
    const real inv_a = (real)1 / a;
    const real inv_rho = (real)1 / rho;

    // ---- phase B: eigen roles:  e^ = w^ - D (Ghat r0 + Lam w^),  h^ = Ghat x~ -------------
    for (int s = 0; s * erole_per_pass < Re; ++s) {
      const int role = s * erole_per_pass + tid / H;
      const bool ract = role < Re;
      const int t = ract ? role / Mg : 0, j = ract ? role - t * Mg : 0;
      real acc = 0;
      const real* r0row = R0 + t * RS;
      for (int ii = hsub; ii < N; ii += H) acc += Ght[ii * MgP + j] * r0row[ii];
      if (H >= 2) acc += shfl_xor_r<real>(acc, 1);
      if (H >= 4) acc += shfl_xor_r<real>(acc, 2);
      real wh = 0;
      const real* wrow = W + t * MgP;
      for (int r = 0; r < Mg; ++r) wh += Qa[r * MgP + j] * wrow[r];
      const real lj = Lam[j];
      const real gh = acc + lj * wh;
      const real ch = rho * gh / (a + rho * lj);
      const real eh = wh - ch;
      if (ract && hsub == 0) {
        E[t * MgP + j] = eh;
        Hh[t * MgP + j] = (acc + lj * eh) * inv_a;
      }
    }
    __syncthreads();

    const bool check = (it % A.check_every == 0) || it >= A.max_iter;

    // ---- phase C, variable role: x~, relaxation, projection onto B, y1 -------------------
    real zh[TPT];
    {
      real xt[TPT];
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) xt[tt] = r0v[tt];
      if (vact) {
        for (int j = 0; j < Mg; ++j) {
          const real g = Ght[ev * MgP + j];
#pragma unroll
          for (int tt = 0; tt < TPT; ++tt) xt[tt] += g * E[(t0 + tt) * MgP + j];
        }
      }
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) {
        const real xn = xt[tt] * inv_a;
        zh[tt] = alpha * xn + ((real)1 - alpha) * z1[tt] + y1[tt] * inv_rho;
        x[tt] = alpha * xn + ((real)1 - alpha) * x[tt];
        z1[tt] = fmin(fmax(zh[tt], lbv[tt]), ubv[tt]);
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k < A.K) {   // block-uniform
        const unsigned wm = wmask[k];
        real sl = 0, lo_l = Tol<real>::big, hi_l = -Tol<real>::big;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
          if ((wm >> tt) & 1u) {
            sl += z1[tt];
            lo_l = fmin(lo_l, zh[tt] - ubv[tt]);
            hi_l = fmax(hi_l, zh[tt] - lbv[tt]);
          }
        }
        const real s0 = quad_sum<real>(sl);
        real lo = quad_min<real>(lo_l), hi = quad_max<real>(hi_l);
        const real cap = scap[k];
        const real tol = Tol<real>::proj * fmax((real)1, fabs(cap));
        bool need = shas[k] && (eq ? fabs(s0 - cap) > tol : s0 > cap + tol);
        int mode = 0;   // 1: z = clip(zh - m), 2: z = ub, 3: z = lb
        if (need) {
          if (cap >= shi[k]) { mode = 2; need = false; }
          else if (cap <= slo[k]) { mode = 3; need = false; }
          else mode = 1;
        }
        if (!eq) lo = fmax(lo, (real)0);
        real m = fmin(fmax(mu[k], lo), hi);
        int guard = 0;
        while (__any(need)) {
          real gl = 0, nl = 0;
#pragma unroll
          for (int tt = 0; tt < TPT; ++tt) {
            if ((wm >> tt) & 1u) {
              const real v = zh[tt] - m;
              gl += fmin(fmax(v, lbv[tt]), ubv[tt]);
              nl += (v > lbv[tt] && v < ubv[tt]) ? (real)1 : (real)0;
            }
          }
          const real g = quad_sum<real>(gl), nf = quad_sum<real>(nl);
          if (need) {
            const real d = g - cap;
            if (fabs(d) <= tol || ++guard > 80) {
              need = false;
            } else {
              if (d > 0) lo = m; else hi = m;
              real mn = nf > 0 ? m + d / nf : (real)0.5 * (lo + hi);
              if (!(mn > lo && mn < hi)) mn = (real)0.5 * (lo + hi);
              m = mn;
            }
          }
        }
        if (mode == 1) mu[k] = m; else if (shas[k]) mu[k] = 0;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
          if ((wm >> tt) & 1u) {
            if (mode == 1) z1[tt] = fmin(fmax(zh[tt] - m, lbv[tt]), ubv[tt]);
            else if (mode == 2) z1[tt] = ubv[tt];
            else if (mode == 3) z1[tt] = lbv[tt];
          }
        }
      }
    }
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) y1[tt] = rho * (zh[tt] - z1[tt]);

    // ---- phase C, constraint role: G x~ = Q h^, relaxation, projection onto C, y2 ---------
    // State (Z2, Y2, GX) lives in LDS; each (row, t) entry is touched by exactly one thread.
#pragma unroll 1
    for (int role = tid; role < Rc; role += kThreads) {
      const int t = role / Mc, c = role - t * Mc;
      const bool is_peak = A.has_peak && c == Mc - 1;
      const bool pair = soc && !is_peak;
      const int ra = is_peak ? Mg - 1 : c;
      const int rb = pair ? c + M : ra;
      const real* hrow = Hh + t * MgP;
      real zta = 0, ztb = 0;
      for (int k = 0; k < Mg; ++k) {
        const real hv = hrow[k];
        zta += Qb[k * MgP + ra] * hv;
        ztb += Qb[k * MgP + rb] * hv;
      }
      const int ia = t * MgP + ra, ib = t * MgP + rb;
      GX[ia] = alpha * zta + ((real)1 - alpha) * GX[ia];
      const real zha = alpha * zta + ((real)1 - alpha) * Z2[ia] + Y2[ia] * inv_rho;
      if (pair) {
        GX[ib] = alpha * ztb + ((real)1 - alpha) * GX[ib];
        const real zhb = alpha * ztb + ((real)1 - alpha) * Z2[ib] + Y2[ib] * inv_rho;
        const real nrm = sqrt(zha * zha + zhb * zhb);
        const real lim = Lim[c];
        const real sc = nrm > lim ? lim / nrm : (real)1;
        const real za = zha * sc, zb = zhb * sc;
        Y2[ia] = rho * (zha - za); Z2[ia] = za;
        Y2[ib] = rho * (zhb - zb); Z2[ib] = zb;
      } else {
        const real lim = is_peak ? Pk[t] : Lim[c];
        const real za = fmin(zha, lim);
        Y2[ia] = rho * (zha - za); Z2[ia] = za;
      }
    }

    // ---- residuals, termination, rho adaptation (block-uniform decisions) -----------------
    if (check) {
      __syncthreads();   // Y2 / Z2 / GX of every role visible
      real v[6];   // pri, dua, |Ax| |z|, |Px|, |A'y|
      v[0] = v[1] = v[2] = v[3] = v[4] = v[5] = 0;
      if (vact) {
        real gty[TPT];
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) gty[tt] = 0;
        for (int j = 0; j < Mg; ++j) {
          const real g = Gt[ev * MgP + j];
#pragma unroll
          for (int tt = 0; tt < TPT; ++tt) gty[tt] += g * Y2[(t0 + tt) * MgP + j];
        }
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
          v[0] = fmax(v[0], fabs(x[tt] - z1[tt]));
          v[1] = fmax(v[1], fabs(pd * x[tt] + qv[tt] + y1[tt] + gty[tt]));
          v[2] = fmax(v[2], fmax(fabs(x[tt]), fabs(z1[tt])));
          v[4] = fmax(v[4], fabs(pd * x[tt]));
          v[5] = fmax(v[5], fabs(y1[tt] + gty[tt]));
        }
      }
      for (int idx = tid; idx < Tm * Mg; idx += kThreads) {
        const int t = idx / Mg, r = idx - t * Mg;
        const real gxv = GX[t * MgP + r], zv = Z2[t * MgP + r];
        v[0] = fmax(v[0], fabs(gxv - zv));
        v[2] = fmax(v[2], fmax(fabs(gxv), fabs(zv)));
      }
      block_max<real, 6>(v, Red, lane, wave);
      pri = v[0];
      dua = v[1];
      const real npri = v[2];
      const real ndua = fmax(fmax(v[4], v[5]), qnorm);
      const real eps_p = (real)A.eps_abs + (real)A.eps_rel * npri;
      const real eps_d = (real)A.eps_abs + (real)A.eps_rel * ndua;
      if (pri <= eps_p && dua <= eps_d) { status = 1; done = true; }
      else if (it >= A.max_iter) { done = true; }
      else if (A.adapt_every > 0 && it % A.adapt_every == 0) {
        const real sp = pri / fmax(npri, (real)1e-12);
        const real sd = dua / fmax(ndua, (real)1e-12);
        const real ratio = sqrt(sp / fmax(sd, (real)1e-30));
        if (ratio > (real)A.adapt_tol || ratio < (real)1 / (real)A.adapt_tol) {
          rho = fmin(fmax(rho * ratio, (real)1e-6), (real)1e6);
        }
      }
    }

    // ---- next iteration's right-hand sides (with the possibly new rho) --------------------
    if (!done) {
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) {
        r0v[tt] = sigma * x[tt] - qv[tt] + rho * z1[tt] - y1[tt];
        if (vact) R0[(t0 + tt) * RS + ev] = r0v[tt];
      }
      // owner-mapped (same role -> entry map as the projection above): no barrier needed
#pragma unroll 1
      for (int role = tid; role < Rc; role += kThreads) {
        const int t = role / Mc, c = role - t * Mc;
        const bool is_peak = A.has_peak && c == Mc - 1;
        const bool pair = soc && !is_peak;
        const int ia = t * MgP + (is_peak ? Mg - 1 : c);
        W[ia] = rho * Z2[ia] - Y2[ia];
        if (pair) W[ia + M] = rho * Z2[ia + M] - Y2[ia + M];
      }
      __syncthreads();
    }
  }

  // ---- results: the feasible iterate z1 is the schedule ------------------------------------
  real ol = 0;
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int t = t0 + tt;
    if (vact && t < Tm) {
      A.x[((size_t)b * N + ev) * Tm + t] = (double)z1[tt];
      ol += ((real)0.5 * pd_user * z1[tt] + qv[tt]) * z1[tt];
    }
  }
  ol = wave_sum<real>(ol);
  __syncthreads();
  if (lane == 0) Red[wave * kNumRed] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < kWaves; ++wv) o += Red[wv * kNumRed];
    A.status[b] = status;
    A.iters[b] = it;
    A.pri[b] = (double)pri;
    A.dua[b] = (double)dua;
    A.obj[b] = (double)o;
  }
}

}  // namespace acnqp
