// General-shape fallback of the batched MPC QP solver: any horizon, N <= 1024, state streamed
// through a per-problem global-memory workspace (L2 resident) instead of registers.
//
// Same ADMM as acn_qp_tiled.hpp (see there for the algorithm); this kernel trades speed for
// generality so that the reference's large scenarios (N = 54, T = 144 stress tests, t_aco.py:286-466,
// and the offline algorithm, adacharge.py:196-294) run through the same C ABI.  One workgroup of 256 / 512 /
// 1024 threads per problem (by size, acn_qp_api.hip); plain loops; one thread per session for the water-filling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acn_qp_tiled.hpp"

namespace acnqp {

constexpr int kGenThreadsMax = 1024;   // workgroup sizes 256 / 512 / 1024, chosen by problem size (acn_qp_api.hip)
constexpr int kGenAccelMax = 5;   // Anderson ring slots (the tiled kernel's number)

struct GeneralArgs {
  TiledArgs t;          // same site / problem / result / option fields as the tiled kernel
  void* work;           // [B][ws_per_problem] reals
  long long ws_per_problem;
  int pair_stride;      // register distance of a SOC pair in the internal row order (4: f64, 1: f32)
};

template <typename real, int kGenThreads>
__device__ inline real block_reduce_max(real v, real* red, int tid) {
  v = wave_max<real>(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  real m = red[0];
  for (int w = 1; w < kGenThreads / 64; ++w) m = fmax(m, red[w]);
  return m;
}

template <typename real, int kGenThreads>
__device__ inline real block_reduce_sum(real v, real* red, int tid) {
  v = wave_sum<real>(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  real m = red[0];
  for (int w = 1; w < kGenThreads / 64; ++w) m += red[w];
  return m;
}

template <typename real, int kGenThreads>
__global__ __launch_bounds__(kGenThreads) void admm_general_kernel(const GeneralArgs SA_kernarg) {
  using M = Mfma<real>;
  __shared__ real red[kGenThreads / 64];
  __shared__ real aaH[kGenAccelMax * kGenAccelMax + kGenAccelMax];
  __shared__ real aaG[kGenAccelMax];
  // passes: pass 0 as the options state it, then cold fixed-penalty retries of a stalled problem (retry_wanted,
  // acn_qp_tiled.hpp).  The WHOLE body is the pass, with the thread / block ids opaque and the argument block read
  // through a per-pass opaque pointer to the kernarg segment: nothing of a pass is invariant across passes, so no
  // pass-invariant address, predicate or argument is kept alive across the solver loop.
  __shared__ int q_slot;
  for (int q_round = 0;; ++q_round) {   // work queue: this workgroup's next problem (queue_next, acn_qp_tiled.hpp)
  const int q_pos = queue_next(SA_kernarg.t.queue, queue_length(SA_kernarg.t), q_round, &q_slot);
  if (q_pos < 0) break;
  int it_total = 0, best_status = 0;
  for (int pass = 0;; ++pass) {
  typedef const __attribute__((address_space(4))) GeneralArgs* KernargP;
  KernargP SAp = (KernargP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(SAp));
  const auto& GA = *SAp;
  const auto& A = GA.t;
  int b_ = q_pos, tid = threadIdx.x;
  asm volatile("" : "+v"(b_));
  asm volatile("" : "+v"(tid));
  const int wg_ = __builtin_amdgcn_readfirstlane(b_);
  const int b = __builtin_amdgcn_readfirstlane(A.order ? A.order[wg_] : wg_);   // the problem this workgroup solves (a uniform value: the load alone would make it a vector register)
  const int max_iter_p = pass == 0 ? A.max_iter : min(A.max_iter, A.retry_max_iter);
  const int adapt_p = pass == 0 ? A.adapt_every : 0;
  const int N = A.N, T = A.Tm, NP = A.NP, MR = A.MR, K = A.K;
  const int n = N * T, mt = MR * T;
  // (the workspace belongs to the workgroup slot when the launch runs off the queue: acn_qp_stream.hpp)
  int ws_slot_ = (int)blockIdx.x;   // (opaque per pass: acn_qp_stream.hpp)
  asm volatile("" : "+v"(ws_slot_));
  ws_slot_ = __builtin_amdgcn_readfirstlane(ws_slot_);
  real* W0 = static_cast<real*>(GA.work) + (size_t)(A.ws_by_slot ? ws_slot_ : b) * GA.ws_per_problem;
  real *x = W0, *z1 = x + n, *y1 = z1 + n, *r0 = y1 + n, *zh = r0 + n, *ub = zh + n;
  real *z2 = ub + n, *y2 = z2 + mt, *gx = y2 + mt, *w = gx + mt, *eh = w + mt, *hh = eh + mt, *zh2 = hh + mt;
  real* mu = zh2 + mt;             // [K*N]
  real* slo = mu + K * N;
  real* shi = slo + K * N;
  real* y1p = shi + K * N;         // duals at the previous residual check (infeasibility certificate)
  real* y2p = y1p + n;
  // Anderson acceleration (same algorithm as acn_qp_tiled.hpp, state in the workspace): u, f of the previous event,
  // the correction applied then (float), the dF / dG rings (float)
  const int D = n + mt;
  const int aa_m = min(A.accel_mem, kGenAccelMax);
  real* uprev = y2p + mt;
  real* fprev = uprev + D;
  float* cprev = reinterpret_cast<float*>(fprev + D);
  float* ringF = cprev + D;
  float* ringG = ringF + (size_t)aa_m * D;
  const real* Gm = static_cast<const real*>(A.G);
  const real* Gh = static_cast<const real*>(A.Ghat);
  const real* Qm = static_cast<const real*>(A.Q);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);
  const double* lbg = A.lb + (size_t)b * n;
  const double* ubg = A.ub + (size_t)b * n;
  const double* qg = A.q + (size_t)b * n;
  const bool eq = A.s_eq[b] != 0;

  real qn = 0, um = 0;
  for (int k = tid; k < n; k += kGenThreads) {
    const real l = (real)lbg[k];
    real u = (real)ubg[k];
    if (u < l) u = l;
    ub[k] = u;
    x[k] = 0; z1[k] = 0; y1[k] = 0;
    r0[k] = -(real)qg[k];
    qn = fmax(qn, fabs((real)qg[k]));
    um = fmax(um, u);
  }
  for (int k = tid; k < mt; k += kGenThreads) { z2[k] = 0; y2[k] = 0; gx[k] = 0; w[k] = 0; }
  real bad = 0;
  for (int s = tid; s < K * N; s += kGenThreads) {
    const int i = s % N;
    const size_t sidx = (size_t)b * K * N + s;
    const int off = A.s_off[sidx], len = A.s_len[sidx];
    real a = 0, c = 0;
    for (int t = off; t < off + len && t < T; ++t) {
      const real l = (real)lbg[i * T + t];
      real u = (real)ubg[i * T + t];
      if (u < l) u = l;
      a += l; c += u;
    }
    mu[s] = 0; slo[s] = a; shi[s] = c;
    if (len > 0) {
      const real cap = (real)A.s_cap[sidx];
      const real slack = (real)64 * M::proj_tol * fmax((real)1, fabs(cap));
      if (a > cap + slack || (eq && c < cap - slack)) bad = 1;
    }
  }
  __syncthreads();
  const real qnorm = block_reduce_max<real, kGenThreads>(qn, red, tid);
  const real ubmax = block_reduce_max<real, kGenThreads>(um, red, tid);
  const real anybad = block_reduce_max<real, kGenThreads>(bad, red, tid);
  const real pd_user = (real)A.pdiag[b];
  const bool dc_on = A.dc != nullptr && (real)A.dc[b] > (real)0;   // block-uniform
  const real pd = effective_pdiag<real>(pd_user, (real)A.reg_rel, qnorm, ubmax, A.horizon[b],
                                        (A.lf != nullptr && (real)A.lf[b] > (real)0) || dc_on);
  if (anybad > 0) {
    for (int k = tid; k < n; k += kGenThreads) A.x[(size_t)b * n + k] = 0;
    if (A.y_out)
      for (int k = tid; k < A.Mg * T; k += kGenThreads) A.y_out[(size_t)b * A.Mg * T + k] = 0;
    if (tid == 0) { A.status[b] = 4; A.iters[b] = 0; A.pri[b] = (double)M::big; A.dua[b] = (double)M::big; A.obj[b] = 0; }
    break;   // (block-uniform) out of the pass loop: the next problem of the queue
  }
  const real sigma = (real)A.sigma, alpha = (real)A.alpha;
  real rho = (real)A.rho0;
  if (pass > 0) {   // fixed penalty retry_rho * 4^(pass - 1)
    rho = (real)A.retry_rho;
    for (int k = 1; k < pass; ++k) rho *= (real)4;
  }
  int status = 2, it = 0, n_adapt = 0, best_it = 0;
  real best_score = M::big;
  real pri = M::big, dua = M::big;
  bool done = false, have_prev = false;
  // ---- energy rows: one thread per session, safeguarded Newton on g(m) = sum clip(zh - m); reads zh and the
  // box-clipped z1, overwrites z1 on the session windows.  Used by the start and by every iteration.
  auto project_sessions = [&]() __attribute__((always_inline)) {
    for (int s = tid; s < K * N; s += kGenThreads) {
      const int i = s % N;
      const size_t sidx = (size_t)b * K * N + s;
      const int off = A.s_off[sidx];
      int len = A.s_len[sidx];
      if (off + len > T) len = T - off;
      if (len <= 0) continue;
      const real cap = (real)A.s_cap[sidx];
      const real* v = zh + i * T + off;
      const double* lb_ = lbg + i * T + off;
      const real* ub_ = ub + i * T + off;
      real* z = z1 + i * T + off;
      real s0 = 0, lo = M::big, hi = -M::big;
      for (int t = 0; t < len; ++t) {
        s0 += z[t];
        lo = fmin(lo, v[t] - ub_[t]);
        hi = fmax(hi, v[t] - (real)lb_[t]);
      }
      const real tol = M::proj_tol * fmax((real)1, fabs(cap));
      const bool need = eq ? fabs(s0 - cap) > tol : s0 > cap + tol;
      if (!need) { mu[s] = 0; continue; }
      if (eq && cap >= shi[s]) { for (int t = 0; t < len; ++t) z[t] = ub_[t]; mu[s] = 0; continue; }
      if (cap <= slo[s]) { for (int t = 0; t < len; ++t) z[t] = (real)lb_[t]; mu[s] = 0; continue; }
      if (!eq && lo < 0) lo = 0;
      real m = fmin(fmax(mu[s], lo), hi);
      for (int guard = 0; guard <= 100; ++guard) {
        real g = 0, nf = 0;
        for (int t = 0; t < len; ++t) {
          const real u = v[t] - m;
          g += fmin(fmax(u, (real)lb_[t]), ub_[t]);
          nf += (u > (real)lb_[t] && u < ub_[t]) ? (real)1 : (real)0;
        }
        const real d = g - cap;
        if (fabs(d) <= tol) break;
        if (d > 0) lo = m; else hi = m;
        real mn = nf > 0 ? m + d / nf : (real)0.5 * (lo + hi);
        if (!(mn > lo && mn < hi)) mn = (real)0.5 * (lo + hi);
        m = mn;
      }
      mu[s] = m;
      for (int t = 0; t < len; ++t) z[t] = fmin(fmax(v[t] - m, (real)lb_[t]), ub_[t]);
    }
  };
  // ---- start: the schedule that ignores the site rows (see acn_qp_tiled.hpp): z1 = Proj_B(-kStartGain q),
  // y1 = -(q + pd z1), z2 = G z1, y2 = 0
  // Warm start (optional; see acn_qp_tiled.hpp): z1 = Proj_B(warm_x), y2 = warm_y (caller's row order and units),
  // y1 = -(q + pd z1 + G' y2)
  const bool warm = pass == 0 && A.warm_x != nullptr && A.warm_y != nullptr;
  for (int k = tid; k < n; k += kGenThreads) {
    zh[k] = warm ? (real)A.warm_x[(size_t)b * n + k] : -(real)kStartGain * (real)qg[k];
    z1[k] = fmin(fmax(zh[k], (real)lbg[k]), ub[k]);
  }
  __syncthreads();
  project_sessions();
  __syncthreads();
  for (int s = tid; s < K * N; s += kGenThreads) mu[s] = 0;
  for (int k = tid; k < mt; k += kGenThreads) {
    const int r = k / T, t = k - r * T;
    real acc = 0;
    for (int i = 0; i < N; ++i) acc += Gm[(size_t)r * NP + i] * z1[i * T + t];
    real yv = 0;
    if (warm) {
      const int ja = A.rowabi[r];
      if (ja >= 0) yv = (real)A.warm_y[((size_t)b * A.Mg + ja) * T + t] / static_cast<const real*>(A.rowscale)[r];
    }
    z2[k] = acc; gx[k] = acc; y2[k] = yv; w[k] = rho * acc - yv;
    if (aa_m > 0) { uprev[n + k] = acc + yv / rho; fprev[n + k] = 0; cprev[n + k] = 0.f; }
  }
  __syncthreads();
  for (int k = tid; k < n; k += kGenThreads) {
    const int i = k / T, t = k - i * T;
    real gty = 0;
    if (warm)
      for (int j = 0; j < MR; ++j) gty += Gm[(size_t)j * NP + i] * y2[j * T + t];
    x[k] = z1[k];
    y1[k] = -((real)qg[k] + pd * z1[k] + gty);
    r0[k] = sigma * x[k] - (real)qg[k] + rho * z1[k] - y1[k];
  }
  if (aa_m > 0) {
    for (int k = tid; k < n; k += kGenThreads) { uprev[k] = z1[k] + y1[k] / rho; fprev[k] = 0; cprev[k] = 0.f; }
    for (int k = tid; k < 2 * aa_m * D; k += kGenThreads) ringF[k] = 0.f;   // dF and dG rings are contiguous
    for (int k = tid; k < kGenAccelMax * kGenAccelMax + kGenAccelMax; k += kGenThreads) aaH[k] = 0;
  }
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  __syncthreads();
  while (!done) {
    ++it;
    const real a = sigma + pd + rho, inv_a = (real)1 / a, inv_rho = (real)1 / rho;
    // ---- eigen space: e^ and h^ -------------------------------------------------------------
    for (int k = tid; k < mt; k += kGenThreads) {
      const int j = k / T, t = k - j * T;
      real acc = 0, whj = 0;
      for (int i = 0; i < N; ++i) acc += Gh[(size_t)j * NP + i] * r0[i * T + t];
      for (int r = 0; r < MR; ++r) whj += Qm[(size_t)r * MR + j] * w[r * T + t];
      const real lj = Lm[j];
      const real e_ = whj - (rho / (a + rho * lj)) * (acc + lj * whj);
      eh[k] = e_;
      hh[k] = (acc + lj * e_) * inv_a;
    }
    __syncthreads();
    // ---- x~, relaxation, box clip;  G x~, relaxation of the site rows ---------------------------
    for (int k = tid; k < n; k += kGenThreads) {
      const int i = k / T, t = k - i * T;
      real v = r0[k];
      for (int j = 0; j < MR; ++j) v += Gh[(size_t)j * NP + i] * eh[j * T + t];
      const real xn = v * inv_a;
      const real zz = alpha * xn + ((real)1 - alpha) * z1[k] + y1[k] * inv_rho;
      zh[k] = zz;
      x[k] = alpha * xn + ((real)1 - alpha) * x[k];
      z1[k] = fmin(fmax(zz, (real)lbg[k]), ub[k]);
    }
    for (int k = tid; k < mt; k += kGenThreads) {
      const int r = k / T, t = k - r * T;
      real zt = 0;
      for (int j = 0; j < MR; ++j) zt += Qm[(size_t)r * MR + j] * hh[j * T + t];
      gx[k] = alpha * zt + ((real)1 - alpha) * gx[k];
      zh2[k] = alpha * zt + ((real)1 - alpha) * z2[k] + y2[k] * inv_rho;
    }
    __syncthreads();
    const bool check = (it % A.check_every == 0) || it >= max_iter_p;
    // ---- Anderson acceleration event (see acn_qp_tiled.hpp; u = (zh, zh2), block-uniform control flow) --------
    if (aa_m > 0 && it % kAaPeriod == 0) {
      auto g_at = [&](int k) -> real& { return k < n ? zh[k] : zh2[k - n]; };
      const bool col = aa_have_prev;
      const int slot = aa_head;
      float* cF = ringF + (size_t)slot * D;
      float* cG = ringG + (size_t)slot * D;
      real part[kGenAccelMax + 2];
      for (int j = 0; j < kGenAccelMax + 2; ++j) part[j] = 0;
      for (int k = tid; k < D; k += kGenThreads) {
        const real g_ = g_at(k), f = g_ - uprev[k];
        part[kGenAccelMax + 1] += f * f;
        if (col) {   // the new column pair (speculatively: it only counts once marked live)
          cF[k] = (float)(f - fprev[k]);
          cG[k] = (float)(g_ - (uprev[k] + (real)cprev[k]));
        }
      }
      const unsigned vnew = aa_valid | (col ? 1u << slot : 0u);
      if (col) {
        for (int k = tid; k < D; k += kGenThreads) {
          const real cf = (real)cF[k];
          for (int j = 0; j < aa_m; ++j)
            if ((vnew >> j) & 1u) part[j] += cf * (real)ringF[(size_t)j * D + k];
          part[kGenAccelMax] += cf * (g_at(k) - uprev[k]);
        }
      }
      real d[kGenAccelMax + 2];
      for (int j = 0; j < kGenAccelMax + 2; ++j) d[j] = block_reduce_sum<real, kGenThreads>(part[j], red, tid);
      const real fn = sqrt(d[kGenAccelMax + 1]);
      bool keep = col;
      if (aa_was && fn > (real)kAaSafe * fn_prev) {   // the accelerated step made things worse: clear, back off
        aa_cnt = 0; aa_head = 0; aa_valid = 0; keep = false;
        __syncthreads();
        for (int k = tid; k < kGenAccelMax * kGenAccelMax + kGenAccelMax; k += kGenThreads) aaH[k] = 0;
        aa_cool = aa_pen;
        aa_pen = aa_pen < 64 ? 2 * aa_pen : 64;
      } else if (aa_cool > 0) --aa_cool;
      __syncthreads();
      if (keep) {
        aa_valid |= 1u << slot;
        if (tid == 0) {
          for (int j = 0; j < aa_m; ++j) {
            if (!((aa_valid >> j) & 1u)) continue;
            aaH[slot * kGenAccelMax + j] = d[j];
            aaH[j * kGenAccelMax + slot] = d[j];
            if (j != slot) aaH[kGenAccelMax * kGenAccelMax + j] += d[j];
          }
          aaH[kGenAccelMax * kGenAccelMax + slot] = d[kGenAccelMax];
        }
        aa_head = slot + 1 == aa_m ? 0 : slot + 1;
        aa_cnt = aa_cnt < aa_m ? aa_cnt + 1 : aa_m;
      }
      for (int k = tid; k < D; k += kGenThreads) fprev[k] = g_at(k) - uprev[k];
      aa_have_prev = true; fn_prev = fn; aa_was = false;
      __syncthreads();
      real dself = 0;   // |dF_new|^2
      for (int j = 0; j < kGenAccelMax; ++j) dself = j == slot ? d[j] : dself;
      // no extrapolation while the map drifts (|dF_new| <= kAaDrift |f|)
      const bool apply = aa_cnt > 0 && aa_cool == 0 && !check && dself > (real)(kAaDrift * kAaDrift) * d[kGenAccelMax + 1];
      if (apply) {
        if (tid == 0) {   // gamma = (H + eta I)^-1 b, Gauss-Jordan in the order of the C port
          real Aug[kGenAccelMax][kGenAccelMax + 1];
          real tr = 0;
          for (int i = 0; i < aa_m; ++i) if ((aa_valid >> i) & 1u) tr += aaH[i * kGenAccelMax + i];
          const real eta = (real)kAaReg * tr + (real)(sizeof(real) == 8 ? 1e-300 : 1e-37);
          for (int i = 0; i < aa_m; ++i) {
            const bool vi = (aa_valid >> i) & 1u;
            for (int j = 0; j < aa_m; ++j) Aug[i][j] = (vi && ((aa_valid >> j) & 1u)) ? aaH[i * kGenAccelMax + j] : (real)0;
            Aug[i][i] = vi ? Aug[i][i] + eta : (real)1;
            Aug[i][aa_m] = vi ? aaH[kGenAccelMax * kGenAccelMax + i] : (real)0;
          }
          for (int k = 0; k < aa_m; ++k) {
            const real inv = (real)1 / Aug[k][k];
            real rs[kGenAccelMax + 1], ck[kGenAccelMax];
            for (int j = 0; j <= aa_m; ++j) rs[j] = Aug[k][j] * inv;
            for (int i = 0; i < aa_m; ++i) ck[i] = Aug[i][k];
            for (int i = 0; i < aa_m; ++i)
              for (int j = 0; j <= aa_m; ++j) Aug[i][j] = i == k ? rs[j] : Aug[i][j] - ck[i] * rs[j];
          }
          for (int i = 0; i < aa_m; ++i) aaG[i] = Aug[i][aa_m];
        }
        __syncthreads();
      }
      for (int k = tid; k < D; k += kGenThreads) {
        real cor = 0;
        if (apply)
          for (int j = 0; j < aa_m; ++j)
            if ((aa_valid >> j) & 1u) cor += aaG[j] * (real)ringG[(size_t)j * D + k];
        const float c_ = (float)cor;            // kept as float; u = g - c with exactly that c
        real& g_ = g_at(k);
        g_ -= (real)c_;
        uprev[k] = g_;
        cprev[k] = c_;
        if (k < n) z1[k] = fmin(fmax(g_, (real)lbg[k]), ub[k]);
      }
      aa_was = apply;
      __syncthreads();
    }
    project_sessions();
    // ---- site rows: projection (internal row order: a SOC pair is `pair_stride` rows apart) -------
    for (int k = tid; k < mt; k += kGenThreads) {
      const int r = k / T, t = k - r * T;
      const int ty = A.rowtype[r];
      if (ty == kRowMax && dc_on) continue;   // written by wave 0 below, and by nobody else (no write-write race)
      real zn = zh2[k];
      if (ty == kRowBox) zn = fmin(zn, RL[r]);
      else if (ty == kRowQuad) zn = zn * (rho / (rho + (A.lf ? (real)(A.lf[b] / (A.flat_scale * A.flat_scale)) : (real)0)));
      else if (ty == kRowPeak) {
        const double pv = A.peak ? A.peak[(size_t)b * T + t] : 1e300;
        zn = fmin(zn, pv < (double)M::big ? (real)(pv * A.peak_scale) : M::big);
      } else if (ty == kRowSocRe || ty == kRowSocIm) {
        const int rr = ty == kRowSocRe ? r : r - GA.pair_stride;
        const real re = zh2[rr * T + t], im = zh2[(rr + GA.pair_stride) * T + t];
        const real n2 = re * re + im * im, lim = RL[rr];
        if (n2 > lim * lim) zn = zn * (lim / sqrt(n2));
      }
      y2[k] = rho * (zh2[k] - zn);
      z2[k] = zn;
    }
    // demand charge: horizon-wide prox on the "max" row (see acn_qp_tiled.hpp), by wave 0 alone: lanes stride over
    // the periods, the reductions leave identical values in every lane, so the Newton iteration is wave-uniform.
    // The strided loop above skips this row, so these are the only writes to its y2 / z2 entries.
    if (dc_on && tid < 64) {
      for (int r = 0; r < MR; ++r)
        if (A.rowtype[r] == kRowMax) {
          const real cw = (real)(A.dc[b] / A.max_scale) * inv_rho, fl = A.dfloor ? (real)(A.dfloor[b] * A.max_scale) : (real)0;
          const real* zv = zh2 + r * T;
          real vl = -M::big;
          for (int t = tid; t < T; t += 64) vl = fmax(vl, zv[t]);
          const real vmax = wave_max<real>(vl);
          real tau = vmax - cw;
          for (int guard = 0; guard < 200; ++guard) {
            real Sl = 0, nl = 0;
            for (int t = tid; t < T; t += 64) if (zv[t] > tau) { Sl += zv[t] - tau; nl += 1; }
            const real S = wave_sum<real>(Sl), nn = wave_sum<real>(nl);
            const real f = S - cw;
            const real tn = nn > 0 ? tau + f / nn : vmax - cw;
            if (fabs(f) <= M::proj_tol * fmax((real)1, cw) * (real)16 || tn == tau) break;
            tau = tn;
          }
          const real lev = fmax(tau, fl);
          for (int t = tid; t < T; t += 64) {
            const real zn = fmin(zv[t], lev);
            y2[r * T + t] = rho * (zv[t] - zn);
            z2[r * T + t] = zn;
          }
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += kGenThreads) y1[k] = rho * (zh[k] - z1[k]);
    // ---- residuals, termination, rho adaptation ------------------------------------------------
    if (check) {
      real v0 = 0, v1 = 0, v2 = 0, v4 = 0, v5 = 0;
      for (int k = tid; k < n; k += kGenThreads) {
        const int i = k / T, t = k - i * T;
        real gty = 0;
        for (int j = 0; j < MR; ++j) gty += Gm[(size_t)j * NP + i] * y2[j * T + t];
        const real yk = rho * (zh[k] - z1[k]);
        v0 = fmax(v0, fabs(x[k] - z1[k]));
        v1 = fmax(v1, fabs(pd * x[k] + (real)qg[k] + yk + gty));
        v2 = fmax(v2, fmax(fabs(x[k]), fabs(z1[k])));
        v4 = fmax(v4, fabs(pd * x[k]));
        v5 = fmax(v5, fabs(yk + gty));
      }
      for (int k = tid; k < mt; k += kGenThreads) {
        v0 = fmax(v0, fabs(gx[k] - z2[k]));
        v2 = fmax(v2, fmax(fabs(gx[k]), fabs(z2[k])));
      }
      pri = block_reduce_max<real, kGenThreads>(v0, red, tid);
      dua = block_reduce_max<real, kGenThreads>(v1, red, tid);
      const real npri = block_reduce_max<real, kGenThreads>(v2, red, tid);
      const real ndua = fmax(fmax(block_reduce_max<real, kGenThreads>(v4, red, tid), block_reduce_max<real, kGenThreads>(v5, red, tid)), qnorm);
      if (pri <= (real)A.eps_abs + (real)A.eps_rel * npri && dua <= (real)A.eps_abs + (real)A.eps_rel * ndua) { status = 1; done = true; }
      if (!done && have_prev) {
        // ---- primal infeasibility certificate (see acn_qp_tiled.hpp): v = y - y(previous check) ------------
        real vnl = 0, atl = 0;
        for (int k = tid; k < n; k += kGenThreads) {
          const int i = k / T, t = k - i * T;
          const real v1 = y1[k] - y1p[k];
          real gtv = 0;
          for (int j = 0; j < MR; ++j) gtv += Gm[(size_t)j * NP + i] * (y2[j * T + t] - y2p[j * T + t]);
          vnl = fmax(vnl, fabs(v1));
          atl = fmax(atl, fabs(v1 + gtv));
        }
        for (int k = tid; k < mt; k += kGenThreads) vnl = fmax(vnl, fabs(y2[k] - y2p[k]));
        const real vn = block_reduce_max<real, kGenThreads>(vnl, red, tid);
        const real atv = block_reduce_max<real, kGenThreads>(atl, red, tid);
        const real vtol = (real)1e-4 * vn;
        if (vn > (real)1e-12 * fmax((real)1, qnorm) && atv <= vtol) {   // block-uniform
          real ssum = 0, bad = 0;
          for (int k = tid; k < mt; k += kGenThreads) {
            const int r = k / T, t = k - r * T;
            const int ty = A.rowtype[r];
            const real v2 = y2[k] - y2p[k];
            if (ty == kRowBox) { ssum += RL[r] * fmax(v2, (real)0); if (v2 < -vtol) bad = 1; }
            else if (ty == kRowPeak) {
              const double pv = A.peak ? A.peak[(size_t)b * T + t] : 1e300;
              if (pv < (double)M::big) ssum += (real)(pv * A.peak_scale) * fmax(v2, (real)0); else if (v2 > vtol) bad = 1;
              if (v2 < -vtol) bad = 1;
            } else if (ty == kRowSocRe) {
              const real vi = y2[(r + GA.pair_stride) * T + t] - y2p[(r + GA.pair_stride) * T + t];
              ssum += RL[r] * sqrt(v2 * v2 + vi * vi);
            } else if (ty == kRowSocIm) {
            } else if (fabs(v2) > vtol) bad = 1;   // free / prox rows admit no ray
          }
          // sessions (zh is free at this point: coverage flags of the periods that lie in some window)
          for (int k = tid; k < n; k += kGenThreads) zh[k] = 0;
          __syncthreads();
          for (int s = tid; s < K * N; s += kGenThreads) {
            const int i = s % N;
            const size_t sidx = (size_t)b * K * N + s;
            const int off = A.s_off[sidx];
            int len = A.s_len[sidx];
            if (off + len > T) len = T - off;
            if (len <= 0) continue;
            real lmin = M::big, lmax = -M::big;
            for (int t = off; t < off + len; ++t) {
              const real v1 = y1[i * T + t] - y1p[i * T + t];
              lmin = fmin(lmin, v1); lmax = fmax(lmax, v1);
              zh[i * T + t] = 1;
            }
            real best = M::big;
            for (int j = 0; j < 3; ++j) {
              real l_ = j == 0 ? lmin : (j == 1 ? lmax : (real)0);
              if (!eq) l_ = fmax(l_, (real)0);
              real ph = l_ * (real)A.s_cap[sidx];
              for (int t = off; t < off + len; ++t) {
                const real dv = (y1[i * T + t] - y1p[i * T + t]) - l_;
                ph += ub[i * T + t] * fmax(dv, (real)0) + (real)lbg[i * T + t] * fmin(dv, (real)0);
              }
              best = fmin(best, ph);
            }
            ssum += best;
          }
          __syncthreads();
          for (int k = tid; k < n; k += kGenThreads)
            if (zh[k] == (real)0) ssum += (real)lbg[k] * (y1[k] - y1p[k]);
          const real stot = block_reduce_sum<real, kGenThreads>(ssum, red, tid);
          const real anyb = block_reduce_max<real, kGenThreads>(bad, red, tid);
          if (anyb == (real)0 && stot < -vtol) { status = 3; done = true; }
        }
      }
      if (!done) {   // snapshot for the next certificate test
        // rounded to single precision like the tiled kernel's register snapshot (one certificate rule everywhere)
        for (int k = tid; k < n; k += kGenThreads) y1p[k] = (real)(float)y1[k];
        for (int k = tid; k < mt; k += kGenThreads) y2p[k] = (real)(float)y2[k];
        have_prev = true;
      }
      const real score = fmax(pri / fmax((real)A.eps_abs + (real)A.eps_rel * npri, (real)1e-300),
                              dua / fmax((real)A.eps_abs + (real)A.eps_rel * ndua, (real)1e-300));
      if (score < (real)kStallGain * best_score) { best_score = score; best_it = it; }
      const bool inacc = inaccurate_ok<real>(pri, dua, npri, ndua, A.eps_abs, A.eps_rel, A.inacc_floor);
      const bool stalled = A.stall_iters > 0 && it - best_it >= A.stall_iters && score <= (real)kStallNear * best_score;   // acn_qp_tiled.hpp
      if (done) {
      } else if (it >= max_iter_p || stalled) {
        done = true;
        if (inacc) status = 5;   // solved, inaccurately
      }
      else if (adapt_p > 0 && it % adapt_p == 0) {
        const real sp = pri / fmax(npri, (real)1e-12), sd = dua / fmax(ndua, (real)1e-12);
        const real ratio = sqrt(sp / fmax(sd, (real)1e-30));
        const real tol_eff = (real)A.adapt_tol * ((real)1 + (real)n_adapt * (real)(1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < (real)1 / tol_eff) {
          rho = fmin(fmax(rho * ratio, (real)1e-6), (real)1e6);
          ++n_adapt;
          if (aa_m > 0) {   // the fixed-point map changed: restart the ring from the current (z, y)
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
            __syncthreads();
            for (int k = tid; k < kGenAccelMax * kGenAccelMax + kGenAccelMax; k += kGenThreads) aaH[k] = 0;
            for (int k = tid; k < n; k += kGenThreads) uprev[k] = z1[k] + y1[k] / rho;
            for (int k = tid; k < mt; k += kGenThreads) uprev[n + k] = z2[k] + y2[k] / rho;
          }
        }
      }
    }
    if (!done) {
      __syncthreads();
      for (int k = tid; k < n; k += kGenThreads) r0[k] = sigma * x[k] - (real)qg[k] + rho * z1[k] - y1[k];
      for (int k = tid; k < mt; k += kGenThreads) w[k] = rho * z2[k] - y2[k];
      __syncthreads();
    }
  }
  it_total += it;
  __syncthreads();
  if (pass == 0 || status_rank(status) > status_rank(best_status)) {   // block-uniform: this pass beats the earlier ones
  best_status = status;
  real ol = 0;
  for (int k = tid; k < n; k += kGenThreads) {
    A.x[(size_t)b * n + k] = (double)z1[k];
    ol += ((real)0.5 * pd_user * z1[k] + (real)qg[k]) * z1[k];
  }
  if (A.y_out)   // site-row multipliers in the caller's row order and units
    for (int k = tid; k < mt; k += kGenThreads) {
      const int r = k / T, t = k - r * T;
      const int ja = A.rowabi[r];
      if (ja >= 0) A.y_out[((size_t)b * A.Mg + ja) * T + t] = (double)(y2[k] * static_cast<const real*>(A.rowscale)[r]);
    }
  ol = wave_sum<real>(ol);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < kGenThreads / 64; ++wv) o += red[wv];
    A.status[b] = status; A.pri[b] = (double)pri; A.dua[b] = (double)dua; A.obj[b] = (double)o;
  }
  }
  if (tid == 0) A.iters[b] = it_total;
  if (!retry_wanted(pass, A.retry_passes, status, it, A.stall_iters, A.adapt_every)) break;
  __syncthreads();
  }   // passes
  }   // work queue
}

}  // namespace acnqp
