// Host side of the C ABI declared in include/acn_qp.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "acn_qp.h"
#include "acn_qp_kernel.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int rc, const std::string& msg) {
  g_last_error = msg;
  return rc;
}

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return fail(ACNQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)

// Cyclic Jacobi eigen-decomposition of a small symmetric matrix (n <= kMaxRows).
// a is overwritten; on return lam[k] are eigenvalues and V[r*n + k] the k-th eigenvector.
void jacobi_eigh(int n, std::vector<double>& a, std::vector<double>& lam, std::vector<double>& V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int p = 0; p < n; ++p)
      for (int q = 0; q < n; ++q) (p == q ? diag : off) += a[(size_t)p * n + q] * a[(size_t)p * n + q];
    if (off <= 1e-32 * (diag > 0 ? diag : 1.0)) break;
    for (int p = 0; p < n - 1; ++p) {
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq;
          a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk;
          a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
    }
  }
  lam.resize(n);
  for (int i = 0; i < n; ++i) lam[i] = a[(size_t)i * n + i];
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct acnqp_handle {
  int device = 0;
  int N = 0, M = 0, Mg = 0, Mc = 0, cone = 0, has_peak = 0;
  double *dG = nullptr, *dGhat = nullptr, *dQ = nullptr, *dLam = nullptr, *dLim = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool timed = false;
  hipStream_t stream = nullptr;   // used by the host-buffer entry point
  DevBuf in, out;                 // staging for the host-buffer entry point
};

template <typename real, int TPT, int KS>
static hipError_t launch_one(const acnqp::KernelArgs& a, hipStream_t st) {
  const acnqp::LdsLayout L(a.N, 4 * TPT, a.Mg, a.M);
  const size_t lds = (size_t)L.total * sizeof(real);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&acnqp::admm_kernel<real, TPT, KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((acnqp::admm_kernel<real, TPT, KS>), dim3(a.B), dim3(acnqp::kThreads), lds, st, a);
  return hipGetLastError();
}

template <typename real, int KS>
static hipError_t launch_for_tmax(const acnqp::KernelArgs& a, hipStream_t st) {
  if (a.Tm <= 12) return launch_one<real, 3, KS>(a, st);
  if (a.Tm <= 16) return launch_one<real, 4, KS>(a, st);
  if (a.Tm <= 24) return launch_one<real, 6, KS>(a, st);
  return launch_one<real, 8, KS>(a, st);
}

template <typename real>
static hipError_t launch_for_k(const acnqp::KernelArgs& a, hipStream_t st) {
  if (a.K == 1) return launch_for_tmax<real, 1>(a, st);
  return launch_for_tmax<real, acnqp::kMaxK>(a, st);
}

extern "C" {

int32_t acnqp_abi_version(void) { return ACNQP_ABI_VERSION; }

const char* acnqp_last_error(void) { return g_last_error.c_str(); }

void acnqp_default_options(acnqp_options* o) {
  if (!o) return;
  o->eps_abs = 1e-8;
  o->eps_rel = 1e-8;
  o->max_iter = 20000;
  o->check_every = 10;
  o->adapt_every = 50;
  o->rho = 0.003;
  o->sigma = 1e-6;
  o->alpha = 1.6;
  o->adapt_tol = 5.0;
  o->reg_rel = 5e-3;
  o->precision = 64;
  o->reserved = 0;
}

int acnqp_create(const acnqp_site* site, int32_t device_id, acnqp_handle** out) {
  if (!site || !out) return fail(ACNQP_ERR_INVALID, "acnqp_create: null argument");
  *out = nullptr;
  const int N = site->n_evse, M = site->n_infra, Mg = site->n_rows;
  if (N < 1 || N > acnqp::kMaxEvse)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: n_evse must be in [1, 64] for the resident kernel");
  if (site->cone != ACNQP_CONE_LINEAR && site->cone != ACNQP_CONE_SOC)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: cone must be ACNQP_CONE_LINEAR or ACNQP_CONE_SOC");
  const int expect = (site->cone == ACNQP_CONE_SOC ? 2 * M : M) + (site->has_peak ? 1 : 0);
  if (M < 0 || Mg != expect || Mg > acnqp::kMaxRows)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: n_rows inconsistent with n_infra/cone/has_peak or > 40");
  if (Mg > 0 && !site->G) return fail(ACNQP_ERR_INVALID, "acnqp_create: G is null");
  if (M > 0 && !site->limits) return fail(ACNQP_ERR_INVALID, "acnqp_create: limits is null");
  for (int j = 0; j < M; ++j)
    if (!(site->limits[j] >= 0)) return fail(ACNQP_ERR_INVALID, "acnqp_create: limits must be >= 0");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(ACNQP_ERR_NO_DEVICE, "acnqp_create: no HIP device visible");
  if (device_id < 0 || device_id >= ndev) return fail(ACNQP_ERR_INVALID, "acnqp_create: bad device_id");
  HIP_TRY(hipSetDevice(device_id));

  // eigen-decomposition of G G' and the rotated rows Ghat = Q' G
  std::vector<double> GGt((size_t)Mg * Mg, 0.0), lam, Q;
  for (int r = 0; r < Mg; ++r)
    for (int c = 0; c < Mg; ++c) {
      double s = 0;
      for (int i = 0; i < N; ++i) s += site->G[(size_t)r * N + i] * site->G[(size_t)c * N + i];
      GGt[(size_t)r * Mg + c] = s;
    }
  jacobi_eigh(Mg, GGt, lam, Q);
  double lmax = 0;
  for (int k = 0; k < Mg; ++k) lmax = std::fmax(lmax, lam[k]);
  std::vector<double> Ghat((size_t)Mg * N, 0.0);
  for (int k = 0; k < Mg; ++k) {
    if (lam[k] < 1e-12 * std::fmax(1.0, lmax)) { lam[k] = 0.0; continue; }
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int r = 0; r < Mg; ++r) s += Q[(size_t)r * Mg + k] * site->G[(size_t)r * N + i];
      Ghat[(size_t)k * N + i] = s;
    }
  }

  acnqp_handle* h = new acnqp_handle();
  h->device = device_id;
  h->N = N; h->M = M; h->Mg = Mg; h->cone = site->cone; h->has_peak = site->has_peak ? 1 : 0;
  h->Mc = M + h->has_peak;
  auto up = [&](double** d, const double* src, size_t n) -> hipError_t {
    hipError_t e = hipMalloc((void**)d, (n ? n : 1) * sizeof(double));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(*d, src, n * sizeof(double), hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = up(&h->dG, site->G, (size_t)Mg * N);
  if (e == hipSuccess) e = up(&h->dGhat, Ghat.data(), (size_t)Mg * N);
  if (e == hipSuccess) e = up(&h->dQ, Q.data(), (size_t)Mg * Mg);
  if (e == hipSuccess) e = up(&h->dLam, lam.data(), (size_t)Mg);
  if (e == hipSuccess) e = up(&h->dLim, site->limits, (size_t)M);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_start);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_stop);
  if (e == hipSuccess) e = hipStreamCreate(&h->stream);
  if (e != hipSuccess) {
    std::string msg = std::string("acnqp_create: ") + hipGetErrorString(e);
    acnqp_destroy(h);
    return fail(ACNQP_ERR_HIP, msg);
  }
  *out = h;
  return ACNQP_OK;
}

void acnqp_destroy(acnqp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->dG) (void)hipFree(h->dG);
  if (h->dGhat) (void)hipFree(h->dGhat);
  if (h->dQ) (void)hipFree(h->dQ);
  if (h->dLam) (void)hipFree(h->dLam);
  if (h->dLim) (void)hipFree(h->dLim);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  h->in.release();
  h->out.release();
  delete h;
}

static int check_problem_shapes(const acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o,
                                const acnqp_results* r) {
  if (!h || !p || !o || !r) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null argument");
  if (p->batch < 0) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: negative batch");
  if (p->batch == 0) return ACNQP_OK;
  if (p->t_max < 1 || p->t_max > 32)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: t_max must be in [1, 32] for the resident kernel");
  if (p->k_sessions < 1 || p->k_sessions > acnqp::kMaxK)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: k_sessions must be in [1, 4]");
  if ((long)h->Mc * p->t_max > (long)40 * 32)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: too many (row, period) pairs");
  if (!p->horizon || !p->lb || !p->ub || !p->q || !p->pdiag || !p->s_off || !p->s_len || !p->s_cap || !p->s_eq)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null problem array");
  if (h->has_peak && !p->peak) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: site has a peak row but peak is null");
  if (!r->x || !r->status || !r->iters || !r->pri_res || !r->dua_res || !r->obj)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null result array");
  if (!(o->eps_abs >= 0) || !(o->eps_rel >= 0) || o->max_iter < 1 || o->check_every < 1 || !(o->rho > 0) ||
      !(o->sigma >= 0) || !(o->alpha > 0 && o->alpha < 2) || !(o->adapt_tol > 1) || !(o->reg_rel >= 0) ||
      o->adapt_every < 0)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: invalid option value");
  if (o->precision != 64 && o->precision != 32)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: precision must be 64 or 32");
  return ACNQP_OK;
}

int acnqp_solve_batch_device(acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o, acnqp_results* r,
                             void* hip_stream) {
  int rc = check_problem_shapes(h, p, o, r);
  if (rc != ACNQP_OK) return rc;
  if (p->batch == 0) return ACNQP_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  acnqp::KernelArgs a;
  a.B = p->batch; a.N = h->N; a.Tm = p->t_max; a.K = p->k_sessions;
  a.Mg = h->Mg; a.M = h->M; a.Mc = h->Mc; a.cone = h->cone; a.has_peak = h->has_peak;
  a.G = h->dG; a.Ghat = h->dGhat; a.Q = h->dQ; a.lam = h->dLam; a.limits = h->dLim;
  a.horizon = p->horizon; a.lb = p->lb; a.ub = p->ub; a.q = p->q; a.pdiag = p->pdiag;
  a.s_off = p->s_off; a.s_len = p->s_len; a.s_cap = p->s_cap; a.s_eq = p->s_eq; a.peak = p->peak;
  a.x = r->x; a.status = r->status; a.iters = r->iters; a.pri = r->pri_res; a.dua = r->dua_res; a.obj = r->obj;
  a.eps_abs = o->eps_abs; a.eps_rel = o->eps_rel; a.rho0 = o->rho; a.sigma = o->sigma; a.alpha = o->alpha;
  a.adapt_tol = o->adapt_tol; a.reg_rel = o->reg_rel;
  a.max_iter = o->max_iter; a.check_every = o->check_every; a.adapt_every = o->adapt_every;
  HIP_TRY(hipEventRecord(h->ev_start, st));
  hipError_t e = (o->precision == 32) ? launch_for_k<float>(a, st) : launch_for_k<double>(a, st);
  if (e != hipSuccess) return fail(ACNQP_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  HIP_TRY(hipEventRecord(h->ev_stop, st));
  h->timed = true;
  return ACNQP_OK;
}

float acnqp_last_kernel_ms(acnqp_handle* h) {
  if (!h || !h->timed) return -1.0f;
  if (hipEventSynchronize(h->ev_stop) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, h->ev_start, h->ev_stop) != hipSuccess) return -1.0f;
  return ms;
}

int acnqp_solve_batch(acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o, acnqp_results* r) {
  int rc = check_problem_shapes(h, p, o, r);
  if (rc != ACNQP_OK) return rc;
  if (p->batch == 0) return ACNQP_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = p->batch, N = h->N, Tm = p->t_max, K = p->k_sessions;
  const size_t nv = B * N * Tm, ns = B * K * N;
  // one staging allocation for inputs, one for outputs; 256-byte aligned slices
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_lb = off; off += al(nv * 8);
  const size_t o_ub = off; off += al(nv * 8);
  const size_t o_q = off; off += al(nv * 8);
  const size_t o_pd = off; off += al(B * 8);
  const size_t o_hz = off; off += al(B * 4);
  const size_t o_so = off; off += al(ns * 4);
  const size_t o_sl = off; off += al(ns * 4);
  const size_t o_sc = off; off += al(ns * 8);
  const size_t o_eq = off; off += al(B);
  const size_t o_pk = off; off += al(p->peak ? B * Tm * 8 : 0);
  HIP_TRY(h->in.reserve(off));
  size_t ooff = 0;
  const size_t r_x = ooff; ooff += al(nv * 8);
  const size_t r_st = ooff; ooff += al(B * 4);
  const size_t r_it = ooff; ooff += al(B * 4);
  const size_t r_pr = ooff; ooff += al(B * 8);
  const size_t r_du = ooff; ooff += al(B * 8);
  const size_t r_ob = ooff; ooff += al(B * 8);
  HIP_TRY(h->out.reserve(ooff));
  char* di = static_cast<char*>(h->in.p);
  char* dout = static_cast<char*>(h->out.p);
  hipStream_t st = h->stream;
  HIP_TRY(hipMemcpyAsync(di + o_lb, p->lb, nv * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_ub, p->ub, nv * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_q, p->q, nv * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_pd, p->pdiag, B * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_hz, p->horizon, B * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_so, p->s_off, ns * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_sl, p->s_len, ns * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_sc, p->s_cap, ns * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(di + o_eq, p->s_eq, B, hipMemcpyHostToDevice, st));
  if (p->peak) HIP_TRY(hipMemcpyAsync(di + o_pk, p->peak, B * Tm * 8, hipMemcpyHostToDevice, st));
  acnqp_problems dp = *p;
  dp.lb = reinterpret_cast<const double*>(di + o_lb);
  dp.ub = reinterpret_cast<const double*>(di + o_ub);
  dp.q = reinterpret_cast<const double*>(di + o_q);
  dp.pdiag = reinterpret_cast<const double*>(di + o_pd);
  dp.horizon = reinterpret_cast<const int32_t*>(di + o_hz);
  dp.s_off = reinterpret_cast<const int32_t*>(di + o_so);
  dp.s_len = reinterpret_cast<const int32_t*>(di + o_sl);
  dp.s_cap = reinterpret_cast<const double*>(di + o_sc);
  dp.s_eq = reinterpret_cast<const uint8_t*>(di + o_eq);
  dp.peak = p->peak ? reinterpret_cast<const double*>(di + o_pk) : nullptr;
  acnqp_results dr;
  dr.x = reinterpret_cast<double*>(dout + r_x);
  dr.status = reinterpret_cast<int32_t*>(dout + r_st);
  dr.iters = reinterpret_cast<int32_t*>(dout + r_it);
  dr.pri_res = reinterpret_cast<double*>(dout + r_pr);
  dr.dua_res = reinterpret_cast<double*>(dout + r_du);
  dr.obj = reinterpret_cast<double*>(dout + r_ob);
  rc = acnqp_solve_batch_device(h, &dp, o, &dr, st);
  if (rc != ACNQP_OK) return rc;
  HIP_TRY(hipMemcpyAsync(r->x, dr.x, nv * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(r->status, dr.status, B * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(r->iters, dr.iters, B * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(r->pri_res, dr.pri_res, B * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(r->dua_res, dr.dua_res, B * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(r->obj, dr.obj, B * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ACNQP_OK;
}

}  // extern "C"
