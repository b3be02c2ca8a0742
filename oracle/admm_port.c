/* ORACLE -- test infrastructure only.  Nothing under adacharge_amd/ links this.
 *
 * Scalar C port of the device ADMM (adacharge_amd/csrc/acn_qp_tiled.hpp), same
 * algorithm and same parameters, double precision, one problem per call;
 * oracle/admm_ref.py is its readable numpy twin.  Used (a) as a near-bitwise
 * checker for the HIP kernel and (b) as bench.py's `cpu_baseline` ("port"),
 * run over a bounded sample with one OpenMP thread per problem.
 *
 * It is NOT the reference's algorithm: the reference calls cvxpy -> ECOS
 * (/root/reference/adacharge/adaptive_charging_optimization.py:318), whose
 * interior-point method is restated in oracle/ipm.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int N, Tm, K, Mg, M, cone, has_peak, has_flat, has_max;
  const double *G, *Ghat, *Q, *lam, *limits;  /* G,Ghat [Mg][N]; Q [Mg][Mg] (Q[r][k]) */
} port_site;

typedef struct {
  double eps_abs, eps_rel, rho, sigma, alpha, adapt_tol, reg_rel;
  int max_iter, check_every, adapt_every;
  int accel_mem;   /* Anderson-acceleration memory (columns); 0 = plain ADMM */
  /* acnqp_options of ABI v7 (include/acn_qp.h): stall window (0 = off), retry passes, their iteration limit,
   * the first retry's fixed penalty, the SOLVED_INACCURATE floor */
  int stall_iters, retry_passes, retry_max_iter;
  double retry_rho, inaccurate_floor;
} port_opts;

static double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* projection of v[0..L) onto {lb<=z<=ub, sum z <= cap (== if eq)}; safeguarded Newton on
 * g(mu) = sum clip(v - mu), warm-started at *mu (kernel: the `while (__any(need))` loop) */
static void project_window(int L, const double* v, const double* lb, const double* ub, double cap, int eq,
                           double sum_lb, double sum_ub, double* mu, double* z) {
  double s0 = 0, lo = 1e300, hi = -1e300;
  for (int t = 0; t < L; ++t) {
    z[t] = clip(v[t], lb[t], ub[t]);
    s0 += z[t];
    if (v[t] - ub[t] < lo) lo = v[t] - ub[t];
    if (v[t] - lb[t] > hi) hi = v[t] - lb[t];
  }
  const double tol = 1e-13 * fmax(1.0, fabs(cap));
  int need = eq ? fabs(s0 - cap) > tol : s0 > cap + tol;
  if (!need) { *mu = 0; return; }
  if (cap >= sum_ub) { for (int t = 0; t < L; ++t) z[t] = ub[t]; *mu = 0; return; }
  if (cap <= sum_lb) { for (int t = 0; t < L; ++t) z[t] = lb[t]; *mu = 0; return; }
  if (!eq && lo < 0) lo = 0;
  double m = fmin(fmax(*mu, lo), hi);
  for (int guard = 0; guard <= 80; ++guard) {
    double g = 0, nf = 0;
    for (int t = 0; t < L; ++t) {
      const double w = v[t] - m;
      g += clip(w, lb[t], ub[t]);
      nf += (w > lb[t] && w < ub[t]) ? 1.0 : 0.0;
    }
    const double d = g - cap;
    if (fabs(d) <= tol) break;
    if (d > 0) lo = m; else hi = m;
    double mn = nf > 0 ? m + d / nf : 0.5 * (lo + hi);
    if (!(mn > lo && mn < hi)) mn = 0.5 * (lo + hi);
    m = mn;
  }
  *mu = m;
  for (int t = 0; t < L; ++t) z[t] = clip(v[t] - m, lb[t], ub[t]);
}

/* returns status: 1 solved, 2 max_iter, 4 empty set */

/* ---- Anderson acceleration (type II) of the ADMM fixed-point map, as the tiled HIP kernel does it.
 * State u = (zhat1, zhat2) = the pre-projection points; every kAaPeriod-th iteration is an "event":
 *   f = g - u_prev  (g = the plain ADMM update reached from u_prev, kAaPeriod iterations later)
 *   columns dF = f - f_prev, dG = g - g_prev are kept (as floats) in a ring of accel_mem slots
 *   gamma = argmin |f - dF gamma|^2 + eta |gamma|^2,  eta = kAaReg * trace(dF'dF)
 *   u_next = g - dG gamma      (skipped on residual-check iterations and while cooling down)
 * No extrapolation while |dF_new| <= kAaDrift |f|: the map is drifting (a plateau of the dual), its
 * differences carry rounding noise only and the least-squares problem is singular.
 * Safeguard: if the residual norm grew by more than kAaSafe after an accelerated step the ring is
 * cleared and acceleration pauses for 1, 2, 4, ... 64 events (exponential back-off).            */
#define AA_MAX 16
static const double kStartGain = 1e5;
static const double kAdaptWiden = 8.0;   /* rho adaptation band: adapt_tol (1 + adaptations / kAdaptWiden) */
static const int kAaPeriod = 5;
/* stall rule of the device kernels (acn_qp_tiled.hpp): no 10 % improvement of max(pri / eps_pri, dua / eps_dua) for
 * kStallIters iterations with the score within kStallNear of its best end the problem (SOLVED_INACCURATE if the
 * residuals are good enough for it, MAX_ITER otherwise) */
static const double kStallGain = 0.9, kStallNear = 1.25;
static const double kAaReg = 1e-4, kAaSafe = 1.2, kAaDrift = 1e-3;

/* solve (H + eta I) gamma = b for the valid columns (LDL', no pivoting: H is a Gram matrix) */
static void aa_solve(int m, const double* H, const double* bvec, unsigned valid, double* gam) {
  double d[AA_MAX], y[AA_MAX], Hm[AA_MAX][AA_MAX], bb[AA_MAX];
  double tr = 0;
  for (int j = 0; j < m; ++j) if ((valid >> j) & 1u) tr += H[j * AA_MAX + j];
  const double eta = kAaReg * tr + 1e-300;
  for (int i = 0; i < m; ++i) {
    const int vi = (valid >> i) & 1u;
    bb[i] = vi ? bvec[i] : 0.0;
    for (int j = 0; j < m; ++j) {
      const int vj = (valid >> j) & 1u;
      Hm[i][j] = (vi && vj) ? H[i * AA_MAX + j] : 0.0;
    }
    Hm[i][i] = vi ? Hm[i][i] + eta : 1.0;
  }
  /* Gauss-Jordan on the augmented system [Hm | b], in the device's operation order (there one lane per
   * entry): a_ij <- a_ij - a_ik * (a_kj / a_kk), pivot row <- a_kj / a_kk.  No pivoting: regularised Gram matrix. */
  double A[AA_MAX][AA_MAX + 1];
  for (int i = 0; i < m; ++i) { for (int j = 0; j < m; ++j) A[i][j] = Hm[i][j]; A[i][m] = bb[i]; }
  for (int k = 0; k < m; ++k) {
    const double inv = 1.0 / A[k][k];
    double rs[AA_MAX + 1], ck[AA_MAX];
    for (int j = 0; j <= m; ++j) rs[j] = A[k][j] * inv;
    for (int i = 0; i < m; ++i) ck[i] = A[i][k];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j <= m; ++j) A[i][j] = (i == k) ? rs[j] : A[i][j] - ck[i] * rs[j];
  }
  for (int i = 0; i < m; ++i) gam[i] = A[i][m];
  (void)d; (void)y;
}

static int solve_one(const port_site* S, const port_opts* O, int horizon, const double* lb, const double* ub_in, const double* q,
                     double pdiag_user, double lf, double dc, double dfloor, const int32_t* s_off, const int32_t* s_len, const double* s_cap, int eq,
                     const double* peak, double* xout, int* iters_out, double* pri_out, double* dua_out,
                     double* obj_out, const double* warm_x, const double* warm_y, double* y_out) {
  const int N = S->N, T = S->Tm, Mg = S->Mg, M = S->M, K = S->K;
  const int n = N * T, mt = Mg * T;
  const int D = n + mt, MM = O->accel_mem > AA_MAX ? AA_MAX : (O->accel_mem > 0 ? O->accel_mem : 0);
  double* buf = (double*)calloc((size_t)(8 * n + 8 * mt + K * N * 3 + 4 * T + Mg + 32 + 4 * D), sizeof(double));
  double* yprev = (double*)calloc((size_t)D + 8, sizeof(double));   /* duals at the previous check (certificate) */
  int have_yprev = 0;
  float* hist = (float*)calloc((size_t)2 * (MM > 0 ? MM : 1) * D, sizeof(float));   /* dF ring, then dG ring */
  double aaH[AA_MAX * AA_MAX], aab[AA_MAX];
  memset(aaH, 0, sizeof aaH); memset(aab, 0, sizeof aab);
  int aa_cnt = 0, aa_head = 0, aa_have_prev = 0, aa_was = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  double fn_prev = 0;
  double *x = buf, *z1 = x + n, *y1 = z1 + n, *r0 = y1 + n, *zh = r0 + n, *ub = zh + n, *xt = ub + n, *gty = xt + n;
  double *z2 = gty + n, *y2 = z2 + mt, *gx = y2 + mt, *w = gx + mt, *wh = w + mt, *gh0 = wh + mt, *eh = gh0 + mt,
         *hh = eh + mt;
  double *mu = hh + mt, *slo = mu + K * N, *shi = slo + K * N, *tmpv = shi + K * N, *zmaxrow = tmpv + Mg + 2;
  double *uprev = zmaxrow + T + 2, *gcur = uprev + D, *fprev = gcur + D, *gprev = fprev + D;
  float *hF = hist, *hG = hist + (size_t)(MM > 0 ? MM : 1) * D;
  double qnorm = 0, ubmax = 0;
  for (int k = 0; k < n; ++k) {
    ub[k] = ub_in[k] < lb[k] ? lb[k] : ub_in[k];
    qnorm = fmax(qnorm, fabs(q[k]));
    ubmax = fmax(ubmax, ub[k]);
  }
  int status = 2, it = 0, best_it = 0;
  double best_score = 1e300;
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < N; ++i) {
      const int L = s_len[k * N + i], o = s_off[k * N + i];
      double a = 0, b = 0;
      for (int t = o; t < o + L && t < T; ++t) { a += lb[i * T + t]; b += ub[i * T + t]; }
      slo[k * N + i] = a; shi[k * N + i] = b;
      if (L > 0) {
        const double cap = s_cap[k * N + i], slack = 64 * 1e-13 * fmax(1.0, fabs(cap));
        if (a > cap + slack || (eq && b < cap - slack)) status = 4;
      }
    }
  double pri = 1e300, dua = 1e300;
  if (status == 4) {
    memset(xout, 0, sizeof(double) * n);
    *iters_out = 0; *pri_out = pri; *dua_out = dua; *obj_out = 0;
    free(buf); free(hist); free(yprev);
    return 4;
  }
  double pd = pdiag_user;
  /* Tikhonov floor: LP-like problems only (kRegResolve / effective_pdiag in acn_qp_tiled.hpp) */
  const int has_prox = (S->has_flat && lf > 0) || (S->has_max && dc > 0);
  if (ubmax > 0 && !has_prox && pdiag_user * ubmax <= 1e-6 * qnorm)
    pd = fmax(pd, O->reg_rel * qnorm / (ubmax * (double)(horizon > 1 ? horizon : 1)));
  double rho = O->rho;
  int n_adapt = 0;   /* adaptations made so far: the tolerance band widens with each (no limit cycles) */
  const double sigma = O->sigma, alpha = O->alpha;
  /* Start: the schedule that ignores the site rows, z1 = Proj_B(-kStartGain q) (every session served as its
   * cost vector prefers, inside its bounds and energy row), with the multiplier that makes it stationary,
   * y1 = -(q + pd z1); site rows at z2 = G z1, y2 = 0.  Exact when no site row binds. */
  /* Warm start (optional; acn_qp_tiled.hpp): z1 = Proj_B(warm_x), y2 = warm_y (already in this solver's row scaling),
   * y1 = -(q + pd z1 + G' y2) */
  const int warm = warm_x != 0 && warm_y != 0;
  for (int k = 0; k < n; ++k) { zh[k] = warm ? warm_x[k] : -kStartGain * q[k]; z1[k] = clip(zh[k], lb[k], ub[k]); }
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < N; ++i) {
      const int L = s_len[k * N + i], o = s_off[k * N + i];
      if (L > 0) {
        double m0 = 0;
        project_window(L, zh + i * T + o, lb + i * T + o, ub + i * T + o, s_cap[k * N + i], eq, slo[k * N + i],
                       shi[k * N + i], &m0, z1 + i * T + o);
      }
    }
  if (warm) memcpy(y2, warm_y, sizeof(double) * mt);
  for (int i = 0; i < N; ++i)
    for (int t = 0; t < T; ++t) {
      const int k = i * T + t;
      double g = 0;
      if (warm) for (int j = 0; j < Mg; ++j) g += S->G[j * N + i] * y2[j * T + t];
      x[k] = z1[k]; y1[k] = -(q[k] + pd * z1[k] + g); uprev[k] = z1[k] + y1[k] / rho;
    }
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < Mg; ++r) {
      double a2 = 0;
      for (int i = 0; i < N; ++i) a2 += S->G[r * N + i] * z1[i * T + t];
      gx[r * T + t] = z2[r * T + t] = a2;
      uprev[n + r * T + t] = a2 + y2[r * T + t] / rho;
    }
  for (int k = 0; k < n; ++k) r0[k] = sigma * x[k] - q[k] + rho * z1[k] - y1[k];
  for (int k = 0; k < mt; ++k) w[k] = rho * z2[k] - y2[k];
  for (it = 1; it <= O->max_iter; ++it) {
    const double a = sigma + pd + rho, inv_a = 1.0 / a, inv_rho = 1.0 / rho;
    /* eigen roles */
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < Mg; ++j) {
        double acc = 0, whj = 0;
        for (int i = 0; i < N; ++i) acc += S->Ghat[j * N + i] * r0[i * T + t];
        for (int r = 0; r < Mg; ++r) whj += S->Q[r * Mg + j] * w[r * T + t];
        const double lj = S->lam[j];
        const double gh = acc + lj * whj;
        const double ch = rho * gh / (a + rho * lj);
        eh[j * T + t] = whj - ch;
        hh[j * T + t] = (acc + lj * eh[j * T + t]) * inv_a;
      }
    /* variable role */
    for (int i = 0; i < N; ++i)
      for (int t = 0; t < T; ++t) {
        double v = r0[i * T + t];
        for (int j = 0; j < Mg; ++j) v += S->Ghat[j * N + i] * eh[j * T + t];
        const double xn = v * inv_a;
        const int k = i * T + t;
        zh[k] = alpha * xn + (1 - alpha) * z1[k] + y1[k] * inv_rho;
        x[k] = alpha * xn + (1 - alpha) * x[k];
      }
    /* constraint role: G x~ = Q h^, relaxation; zh2 = pre-projection point of the site rows */
    double* zh2 = gcur + n;
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < Mg; ++r) {
        double zt = 0;
        for (int k = 0; k < Mg; ++k) zt += S->Q[r * Mg + k] * hh[k * T + t];
        gx[r * T + t] = alpha * zt + (1 - alpha) * gx[r * T + t];
        zh2[r * T + t] = alpha * zt + (1 - alpha) * z2[r * T + t] + y2[r * T + t] * inv_rho;
      }
    const int check = (it % O->check_every == 0) || it >= O->max_iter;
    /* ---- Anderson acceleration event ---- */
    if (MM > 0 && it % kAaPeriod == 0) {
      memcpy(gcur, zh, sizeof(double) * n);
      double fn = 0;
      for (int k = 0; k < D; ++k) { const double f = gcur[k] - uprev[k]; fn += f * f; }
      fn = sqrt(fn);
      if (aa_was && fn > kAaSafe * fn_prev) {   /* the accelerated step made things worse: clear, back off */
        aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = 0;
        memset(aaH, 0, sizeof aaH); memset(aab, 0, sizeof aab);
        aa_cool = aa_pen; aa_pen = aa_pen < 64 ? 2 * aa_pen : 64;
      } else if (aa_cool > 0) --aa_cool;
      if (aa_have_prev) {
        const int c = aa_head;
        float* cf = hF + (size_t)D * c; float* cg = hG + (size_t)D * c;
        for (int k = 0; k < D; ++k) { cf[k] = (float)((gcur[k] - uprev[k]) - fprev[k]); cg[k] = (float)(gcur[k] - gprev[k]); }
        aa_valid |= 1u << c;
        for (int j = 0; j < MM; ++j) {
          if (!((aa_valid >> j) & 1u)) continue;
          const float* cj = hF + (size_t)D * j;
          double dsum = 0;
          for (int k = 0; k < D; ++k) dsum += (double)cf[k] * (double)cj[k];
          aaH[c * AA_MAX + j] = aaH[j * AA_MAX + c] = dsum;
          if (j != c) aab[j] += dsum;   /* dF_j . f_k = dF_j . f_(k-1) + dF_j . dF_c */
        }
        double db = 0;
        for (int k = 0; k < D; ++k) db += (double)cf[k] * (gcur[k] - uprev[k]);
        aab[c] = db;
        aa_head = (aa_head + 1) % MM; if (aa_cnt < MM) ++aa_cnt;
      }
      for (int k = 0; k < D; ++k) { fprev[k] = gcur[k] - uprev[k]; gprev[k] = gcur[k]; }
      aa_have_prev = 1; fn_prev = fn; aa_was = 0;
      const int cnew = (aa_head + MM - 1) % MM;   /* the column this event added (aa_cnt > 0 implies one was) */
      if (aa_cnt > 0 && aa_cool == 0 && !check && aaH[cnew * AA_MAX + cnew] > kAaDrift * kAaDrift * fn * fn) {
        double gam[AA_MAX];
        aa_solve(MM, aaH, aab, aa_valid, gam);
        for (int j = 0; j < MM; ++j) {
          if (!((aa_valid >> j) & 1u)) continue;
          const float* cg = hG + (size_t)D * j;
          const double gj = gam[j];
          for (int k = 0; k < D; ++k) gcur[k] -= gj * (double)cg[k];
        }
        aa_was = 1;
      }
      memcpy(uprev, gcur, sizeof(double) * D);
      memcpy(zh, gcur, sizeof(double) * n);
    }
    /* ---- projections: z1 = Proj_B(zh), z2 = Proj_C(zh2), y = rho (pre-projection - projection) ---- */
    for (int k = 0; k < n; ++k) z1[k] = clip(zh[k], lb[k], ub[k]);
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < N; ++i) {
        const int L = s_len[k * N + i], o = s_off[k * N + i];
        if (L > 0)
          project_window(L, zh + i * T + o, lb + i * T + o, ub + i * T + o, s_cap[k * N + i], eq, slo[k * N + i],
                         shi[k * N + i], &mu[k * N + i], z1 + i * T + o);
      }
    for (int k = 0; k < n; ++k) y1[k] = rho * (zh[k] - z1[k]);
    for (int t = 0; t < T; ++t) {
      for (int r = 0; r < Mg; ++r) tmpv[r] = zh2[r * T + t];
      for (int c = 0; c < M; ++c) {
        if (S->cone == 1) {
          const double za = tmpv[c], zb = tmpv[c + M], nrm = sqrt(za * za + zb * zb), lim = S->limits[c];
          const double sc = nrm > lim ? lim / nrm : 1.0;
          z2[c * T + t] = za * sc; z2[(c + M) * T + t] = zb * sc;
          y2[c * T + t] = rho * (za - za * sc); y2[(c + M) * T + t] = rho * (zb - zb * sc);
        } else {
          const double za = fmin(tmpv[c], S->limits[c]);
          y2[c * T + t] = rho * (tmpv[c] - za); z2[c * T + t] = za;
        }
      }
      if (S->has_flat) {   /* prox of 1/2 lf z^2 on the aggregate-power row */
        const int r = Mg - 1 - S->has_peak - S->has_max;
        const double za = tmpv[r] * (rho / (rho + lf));
        y2[r * T + t] = rho * (tmpv[r] - za); z2[r * T + t] = za;
      }
      if (S->has_max) {   /* keep zhat of the demand-charge row; its prox couples all periods */
        const int r = Mg - 1 - S->has_peak;
        zmaxrow[t] = tmpv[r];
        y2[r * T + t] = 0; z2[r * T + t] = tmpv[r];
      }
      if (S->has_peak) {
        const int r = Mg - 1;
        const double lim = peak ? fmin(peak[t], 1e300) : 1e300;
        const double za = fmin(tmpv[r], lim);
        y2[r * T + t] = rho * (tmpv[r] - za); z2[r * T + t] = za;
      }
    }
    if (S->has_max && dc > 0) {   /* prox of dc * max(max_t z_t, floor) on the demand-charge row */
      const int r = Mg - 1 - S->has_peak;
      const double cw = dc * inv_rho;
      double vmax = -1e300;
      for (int t = 0; t < T; ++t) if (zmaxrow[t] > vmax) vmax = zmaxrow[t];
      double tau = vmax - cw;
      for (int guard = 0; guard < 200; ++guard) {
        double Ssum = 0, nn = 0;
        for (int t = 0; t < T; ++t) if (zmaxrow[t] > tau) { Ssum += zmaxrow[t] - tau; nn += 1; }
        const double f = Ssum - cw;
        const double tn = nn > 0 ? tau + f / nn : vmax - cw;
        if (fabs(f) <= 1e-13 * fmax(1.0, cw) * 16 || tn == tau) break;
        tau = tn;
      }
      const double lev = fmax(tau, dfloor);
      for (int t = 0; t < T; ++t) {
        const double zn = fmin(zmaxrow[t], lev);
        y2[r * T + t] = rho * (zmaxrow[t] - zn); z2[r * T + t] = zn;
      }
    }
    int done = 0;
    if (check) {
      double v0 = 0, v1 = 0, v2 = 0, v4 = 0, v5 = 0;
      for (int i = 0; i < N; ++i)
        for (int t = 0; t < T; ++t) {
          double g = 0;
          for (int j = 0; j < Mg; ++j) g += S->G[j * N + i] * y2[j * T + t];
          const int k = i * T + t;
          v0 = fmax(v0, fabs(x[k] - z1[k]));
          v1 = fmax(v1, fabs(pd * x[k] + q[k] + y1[k] + g));
          v2 = fmax(v2, fmax(fabs(x[k]), fabs(z1[k])));
          v4 = fmax(v4, fabs(pd * x[k]));
          v5 = fmax(v5, fabs(y1[k] + g));
        }
      for (int k = 0; k < mt; ++k) {
        v0 = fmax(v0, fabs(gx[k] - z2[k]));
        v2 = fmax(v2, fmax(fabs(gx[k]), fabs(z2[k])));
      }
      pri = v0; dua = v1;
      const double npri = v2, ndua = fmax(fmax(v4, v5), qnorm);
      if (pri <= O->eps_abs + O->eps_rel * npri && dua <= O->eps_abs + O->eps_rel * ndua) { status = 1; done = 1; }
      if (!done && have_yprev) {
        /* ---- primal infeasibility certificate, as the tiled kernel tests it (acn_qp_tiled.hpp): v = y - y(previous
         * check); if A'v ~ 0 and the support function of B x C at v is negative, no point of B x C solves A r = z. */
        double vn = 0, atv = 0;
        for (int i = 0; i < N; ++i)
          for (int t = 0; t < T; ++t) {
            const int k = i * T + t;
            const double v1 = y1[k] - yprev[k];
            double gtv = 0;
            for (int j = 0; j < Mg; ++j) gtv += S->G[j * N + i] * (y2[j * T + t] - yprev[n + j * T + t]);
            vn = fmax(vn, fabs(v1));
            atv = fmax(atv, fabs(v1 + gtv));
          }
        for (int k = 0; k < mt; ++k) vn = fmax(vn, fabs(y2[k] - yprev[n + k]));
        const double vtol = 1e-4 * vn;
        if (vn > 1e-12 * fmax(1.0, qnorm) && atv <= vtol) {
          int bad = 0;
          double ssum = 0;
          for (int t = 0; t < T; ++t)
            for (int r = 0; r < Mg; ++r) {
              const double v2 = y2[r * T + t] - yprev[n + r * T + t];
              if (r < M && S->cone == 1) {          /* disc: radius * |(v_re, v_im)| */
                const double vi = y2[(r + M) * T + t] - yprev[n + (r + M) * T + t];
                ssum += S->limits[r] * sqrt(v2 * v2 + vi * vi);
              } else if (r < 2 * M && S->cone == 1) {
              } else if (r < M) {                    /* box: z <= limit */
                ssum += S->limits[r] * fmax(v2, 0.0);
                if (v2 < -vtol) bad = 1;
              } else if (S->has_peak && r == Mg - 1) {
                const double pk = peak ? peak[t] : 1e300;
                if (pk < 1e300) ssum += pk * fmax(v2, 0.0); else if (v2 > vtol) bad = 1;
                if (v2 < -vtol) bad = 1;
              } else if (fabs(v2) > vtol) bad = 1;   /* prox rows admit no ray */
            }
          /* sessions: phi(l) = l cap + sum_t [ub (v_t - l)+ + lb (v_t - l)-] bounds the support function for any
           * admissible l; evaluated at min v, max v, 0.  Periods outside every window are pinned to lb (= ub). */
          for (int k = 0; k < n; ++k) xt[k] = 0;   /* coverage flags (xt is free here) */
          for (int k = 0; k < K; ++k)
            for (int i = 0; i < N; ++i) {
              const int L = s_len[k * N + i], o = s_off[k * N + i];
              if (L <= 0) continue;
              double lmin = 1e300, lmax = -1e300;
              for (int t = o; t < o + L && t < T; ++t) {
                const double v1 = y1[i * T + t] - yprev[i * T + t];
                lmin = fmin(lmin, v1); lmax = fmax(lmax, v1);
                xt[i * T + t] = 1;
              }
              double best = 1e300;
              const double cand[3] = {lmin, lmax, 0.0};
              for (int j = 0; j < 3; ++j) {
                double l_ = cand[j];
                if (!eq) l_ = fmax(l_, 0.0);
                double ph = l_ * s_cap[k * N + i];
                for (int t = o; t < o + L && t < T; ++t) {
                  const double dv = (y1[i * T + t] - yprev[i * T + t]) - l_;
                  ph += ub[i * T + t] * fmax(dv, 0.0) + lb[i * T + t] * fmin(dv, 0.0);
                }
                best = fmin(best, ph);
              }
              ssum += best;
            }
          for (int k = 0; k < n; ++k)
            if (xt[k] == 0) ssum += lb[k] * (y1[k] - yprev[k]);
          if (!bad && ssum < -vtol) { status = 3; done = 1; }
        }
      }
      if (!done) {
        /* the device keeps this snapshot in single precision (registers): same rounding here */
        for (int k = 0; k < n; ++k) yprev[k] = (double)(float)y1[k];
        for (int k = 0; k < mt; ++k) yprev[n + k] = (double)(float)y2[k];
        have_yprev = 1;
      }
      const double score = fmax(pri / fmax(O->eps_abs + O->eps_rel * npri, 1e-300), dua / fmax(O->eps_abs + O->eps_rel * ndua, 1e-300));
      if (score < kStallGain * best_score) { best_score = score; best_it = it; }
      /* solved, inaccurately: within 100 x the tolerance, or within cvxpy's OSQP default 1e-5, whichever is looser */
      const double ea = fmax(100.0 * O->eps_abs, O->inaccurate_floor), er = fmax(100.0 * O->eps_rel, O->inaccurate_floor);
      const int inacc = pri <= ea + er * npri && dua <= ea + er * ndua;
      const int stalled = O->stall_iters > 0 && it - best_it >= O->stall_iters && score <= kStallNear * best_score;
      if (done) {
      } else if (it >= O->max_iter || stalled) {
        done = 1;
        if (inacc) status = 5;
      }
      else if (O->adapt_every > 0 && it % O->adapt_every == 0) {
        const double sp = pri / fmax(npri, 1e-12), sd = dua / fmax(ndua, 1e-12);
        const double ratio = sqrt(sp / fmax(sd, 1e-30));
        const double tol_eff = O->adapt_tol * (1.0 + n_adapt / kAdaptWiden);
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = fmin(fmax(rho * ratio, 1e-6), 1e6);
          if (MM > 0) {   /* the fixed-point map changed: restart the ring from the current (z, y) */
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = 0; aa_was = 0;
            memset(aaH, 0, sizeof aaH); memset(aab, 0, sizeof aab);
            for (int k = 0; k < n; ++k) uprev[k] = z1[k] + y1[k] / rho;
            for (int k = 0; k < mt; ++k) uprev[n + k] = z2[k] + y2[k] / rho;
          }
        }
      }
    }
    if (done) break;
    for (int k = 0; k < n; ++k) r0[k] = sigma * x[k] - q[k] + rho * z1[k] - y1[k];
    for (int k = 0; k < mt; ++k) w[k] = rho * z2[k] - y2[k];
  }
  if (it > O->max_iter) it = O->max_iter;
  double obj = 0;
  for (int k = 0; k < n; ++k) { xout[k] = z1[k]; obj += (0.5 * pdiag_user * z1[k] + q[k]) * z1[k]; }
  if (y_out) memcpy(y_out, y2, sizeof(double) * mt);
  *iters_out = it; *pri_out = pri; *dua_out = dua; *obj_out = obj;
  free(buf); free(hist); free(yprev);
  return status;
}

/* Batch driver: same array layout as include/acn_qp.h (host pointers); `threads` OpenMP threads. */
int admm_port_solve_batch(const port_site* S, const port_opts* O, int B, const int32_t* horizon, const double* lb, const double* ub,
                          const double* q, const double* pdiag, const double* lf, const double* dc, const double* dfloor, const int32_t* s_off, const int32_t* s_len,
                          const double* s_cap, const uint8_t* s_eq, const double* peak, double* x, int32_t* status,
                          int32_t* iters, double* pri, double* dua, double* obj, int threads,
                          const double* warm_x, const double* warm_y, double* y_out) {
  const size_t nv = (size_t)S->N * S->Tm, ns = (size_t)S->K * S->N;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    int it = 0;
    status[b] = solve_one(S, O, horizon[b], lb + b * nv, ub + b * nv, q + b * nv, pdiag[b], lf ? lf[b] : 0.0, dc ? dc[b] : 0.0, dfloor ? dfloor[b] : 0.0, s_off + b * ns, s_len + b * ns,
                          s_cap + b * ns, s_eq[b] != 0, peak ? peak + (size_t)b * S->Tm : 0, x + b * nv, &it,
                          pri + b, dua + b, obj + b, warm_x ? warm_x + b * nv : 0,
                          warm_y ? warm_y + (size_t)b * S->Mg * S->Tm : 0, y_out ? y_out + (size_t)b * S->Mg * S->Tm : 0);
    int total = it, last_status = status[b], last_it = it;
    /* retry passes of the device kernels (retry_wanted, acn_qp_tiled.hpp): a problem a pass leaves MAX_ITER /
     * SOLVED_INACCURATE after at least stall_iters (3000 if the rule is off) iterations is solved again from a cold
     * start with the fixed penalty retry_rho * 4^(pass - 1); the best pass is kept (SOLVED > SOLVED_INACCURATE >
     * MAX_ITER, the first of equals), iters is the total */
    for (int pass = 0; pass < O->retry_passes && (last_status == 2 || last_status == 5) &&
                       last_it >= (O->stall_iters > 0 ? O->stall_iters : 3000) && O->adapt_every > 0; ++pass) {
      port_opts O2 = *O;
      O2.rho = O->retry_rho;
      for (int k = 0; k < pass; ++k) O2.rho *= 4.0;
      O2.adapt_every = 0;
      O2.max_iter = O->max_iter < O->retry_max_iter ? O->max_iter : O->retry_max_iter;
      double* x2 = (double*)malloc(sizeof(double) * (nv + (size_t)S->Mg * S->Tm));
      double* y2 = x2 + nv;
      double pri2, dua2, obj2;
      int it2 = 0;
      const int st2 = solve_one(S, &O2, horizon[b], lb + b * nv, ub + b * nv, q + b * nv, pdiag[b], lf ? lf[b] : 0.0, dc ? dc[b] : 0.0, dfloor ? dfloor[b] : 0.0,
                                s_off + b * ns, s_len + b * ns, s_cap + b * ns, s_eq[b] != 0, peak ? peak + (size_t)b * S->Tm : 0, x2, &it2,
                                &pri2, &dua2, &obj2, 0, 0, y_out ? y2 : 0);
      total += it2;
      const int rank_new = st2 == 1 ? 3 : (st2 == 5 ? 2 : (st2 == 2 ? 1 : 0));
      const int rank_old = status[b] == 1 ? 3 : (status[b] == 5 ? 2 : (status[b] == 2 ? 1 : 0));
      if (rank_new > rank_old) {
        memcpy(x + b * nv, x2, sizeof(double) * nv);
        if (y_out) memcpy(y_out + (size_t)b * S->Mg * S->Tm, y2, sizeof(double) * S->Mg * S->Tm);
        status[b] = st2; pri[b] = pri2; dua[b] = dua2; obj[b] = obj2;
      }
      free(x2);
      last_status = st2; last_it = it2;
    }
    iters[b] = total;
  }
  return 0;
}

int admm_port_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
