// Host side of the C ABI declared in include/acn_qp.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "acn_qp.h"
#include "acn_qp_launch.hpp"

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

// Cyclic Jacobi eigen-decomposition of a small symmetric matrix (n <= 48).
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

// pinned host staging, grow-only
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

// Site data in the kernel's internal row order, one copy per arithmetic type (the MFMA C-operand
// row map differs between f64 and f32, and SOC pairs must sit in adjacent registers of one lane).
struct SiteDev {
  bool ready = false;
  int MR = 0;                    // padded rows (multiple of 16)
  double peak_scale = 1, flat_scale = 1, max_scale = 1;   // row-equilibration factors of the prox rows
  void *G = nullptr, *Ghat = nullptr, *Q = nullptr, *lam = nullptr, *rowlim = nullptr;
  void *fragG = nullptr, *fragQ = nullptr;   // Ghat / Q in MFMA A-operand fragment order (tiled kernel)
  void *fragG2 = nullptr, *fragQ2 = nullptr; // the same with the four k-slices of a fragment as two adjacent pairs per lane (long-horizon kernel)
  int32_t* rowtype = nullptr;
  int32_t* rowabi = nullptr;     // internal row -> row of acnqp_site.G (-1: padding)
  void* rowscale = nullptr;      // equilibration factor of each internal row
  double *Gabi = nullptr, *limabi = nullptr;   // acnqp_site.G / limits as the caller gave them (the polish kernel works in the caller's units)
  void release() {
    if (Gabi) (void)hipFree(Gabi);
    if (limabi) (void)hipFree(limabi);
    Gabi = limabi = nullptr;
    for (void** p : {&G, &Ghat, &Q, &lam, &rowlim, &fragG, &fragQ, &fragG2, &fragQ2, &rowscale}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (rowtype) (void)hipFree(rowtype);
    if (rowabi) (void)hipFree(rowabi);
    rowtype = nullptr; rowabi = nullptr;
    ready = false;
  }
};

struct acnqp_handle {
  int device = 0;
  int N = 0, M = 0, Mg = 0, cone = 0, has_peak = 0, has_flat = 0, has_max = 0;
  int NW = 4, NP = 64;
  std::vector<double> G, limits;   // host copy in ABI order
  SiteDev dev64;
  static constexpr int kEvRing = 64;               // event pairs of the most recent launches
  hipEvent_t ev_start[kEvRing] = {}, ev_stop[kEvRing] = {};
  long long launches = 0, reported = 0;           // launches recorded / already handed out by acnqp_kernel_times
  long long ordered_launches = 0;                 // launches whose queue order was sorted by session count (acnqp_ordered_launch_count)
  // host-buffer entry points: kSlots pipeline slots, each with its own stream and device staging, so that the
  // H2D copies, the kernel and the D2H copies of successive chunks of a call overlap
  static constexpr int kSlots = 4;
  struct Slot { hipStream_t st = nullptr; DevBuf in, out, tin; } slot[kSlots];   // tin: session-table staging (acnqp_solve_table)
  // pinned mirrors of a call's SMALL per-problem arrays (scalars, session slots; status ... obj): one H2D and one D2H per
  // chunk instead of nine and five per batch of the call -- a stream operation costs tens of microseconds whatever its size,
  // and a step of 64 batches was 960 of them (run_pipeline)
  HostBuf small_in, small_out;
  hipEvent_t h2d_done[kSlots] = {nullptr, nullptr, nullptr, nullptr};   // chunk c's inputs have landed (run_pipeline: the next chunk's copies queue behind them)
  // per launch stream: the kernel workspace (long-horizon, large-site, general-shape kernels) and the launch's small
  // scheduling buffer (queue counter, then sort keys and queue order).  Launches on different streams never share (or
  // regrow) each other's state, and a stream's own launches are ordered by the stream.
  struct Work { hipStream_t st; DevBuf buf; DevBuf ord; DevBuf pol; long long used; hipEvent_t last; };   // pol: the polish's list and multiplier buffer
  std::vector<Work> work;
  long long work_clock = 0;
  // streams of the CALLER (acnqp_solve_batch_device) beyond the handle's own kSlots: a caller that round-robins one
  // handle over more streams than this pays an event wait + free / malloc per call; 0 < env
  // ACNQP_CALLER_STREAMS <= 256 overrides
  static size_t max_caller_workspaces() {
    static const size_t v = [] {
      const char* e = std::getenv("ACNQP_CALLER_STREAMS");
      const long n = e ? std::atol(e) : 0;
      return (size_t)(n > 0 && n <= 256 ? n : 16);
    }();
    return v;
  }
  DevBuf* workspace_for(hipStream_t st) { return &work_for(st)->buf; }
  Work* work_for(hipStream_t st) {
    for (auto& w : work) if (w.st == st) { w.used = ++work_clock; return &w; }
    // a caller that keeps creating streams must not grow the handle without bound: evict the least recently used
    // entry of a caller stream -- after ITS last launch has finished (the event recorded behind it: that stream may be
    // gone by now), not after the whole device has drained
    bool own = false;
    for (auto& sl : slot) own = own || sl.st == st;
    if (!own) {
      size_t callers = 0, lru = work.size();
      for (size_t k = 0; k < work.size(); ++k) {
        bool mine = false;
        for (auto& sl : slot) mine = mine || sl.st == work[k].st;
        if (mine) continue;
        ++callers;
        if (lru == work.size() || work[k].used < work[lru].used) lru = k;
      }
      if (callers >= max_caller_workspaces()) {
        if (work[lru].last) { (void)hipEventSynchronize(work[lru].last); (void)hipEventDestroy(work[lru].last); }
        work[lru].buf.release();
        work[lru].ord.release();
        work[lru].pol.release();
        work.erase(work.begin() + (long)lru);
      }
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    work.push_back(Work{st, DevBuf(), DevBuf(), DevBuf(), ++work_clock, ev});
    return &work.back();
  }
  void release_work() {
    for (auto& w : work) {
      if (w.last) (void)hipEventDestroy(w.last);
      w.buf.release();
      w.ord.release();
      w.pol.release();
    }
    work.clear();
  }
  int cus = 0;   // compute units of the device (the work-queue launches size their grid from it)
  int32_t* pol_stats = nullptr;   // device counters of the polish kernel, summed over the handle's life (acnqp_polish_stats)
};

namespace {

// ---- launch order: longest expected problem first -------------------------------------------------------------------
// A launch of B problems on S resident workgroup slots ends with its slowest slot; the hardware hands workgroups out in
// index order, so the tail is up to one whole problem long (bench workload, 32 problems per slot: natural order 7.0 %
// above the mean slot, `longest first` by the TRUE iteration counts 0.2 %).  The number of sessions of a problem
// predicts its iteration count well enough (rank correlation 0.81 on that workload: the list schedule by it ends 1.8 %
// above the mean): two tiny kernels sort the problems by it, descending, and every solver kernel maps workgroup ->
// problem through the result.  Results do not depend on the order (a workgroup only touches its own problem).
constexpr int kOrderKeys = 1024;
constexpr int kOrderMinBatch = 768;   // fewer problems than ~1.5 x the resident slots: nothing to level
__global__ __launch_bounds__(256) void order_keys_kernel(const int32_t* s_len, int KN, int B, int32_t* keys) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  int c = 0;
  for (int k = lane; k < KN; k += 64) c += s_len[(size_t)b * KN + k] > 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) keys[b] = c < kOrderKeys ? c : kOrderKeys - 1;
}
__global__ __launch_bounds__(kOrderKeys) void order_sort_kernel(const int32_t* keys, int B, int32_t* order) {
  __shared__ int hist[kOrderKeys], scan[kOrderKeys];
  const int tid = threadIdx.x;
  hist[tid] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += kOrderKeys) atomicAdd(&hist[keys[b]], 1);
  __syncthreads();
  // slot r = kOrderKeys - 1 - key (largest key first): exclusive prefix sum over r
  scan[tid] = hist[kOrderKeys - 1 - tid];
  __syncthreads();
  for (int o = 1; o < kOrderKeys; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  hist[kOrderKeys - 1 - tid] = scan[tid] - hist[kOrderKeys - 1 - tid];   // start of this key's run
  __syncthreads();
  for (int b = tid; b < B; b += kOrderKeys) order[atomicAdd(&hist[keys[b]], 1)] = b;
}

// internal row slot of site row (constraint c, component) -- see acn_qp_tiled.hpp
inline int soc_slot(bool f64, int c, int im) {
  return f64 ? 8 * (c / 4) + (c % 4) + 4 * im : 2 * c + im;
}

int build_site_dev(acnqp_handle* h, SiteDev* d) {
  using real = double;
  const bool f64 = true;
  const int N = h->N, M = h->M, NP = h->NP;
  int raw;
  if (h->cone == ACNQP_CONE_SOC) raw = (f64 ? 8 * ((M + 3) / 4) : 2 * M) + h->has_peak + h->has_flat + h->has_max;
  else raw = M + h->has_peak + h->has_flat + h->has_max;
  const int MR = 16 * ((raw + 15) / 16 > 0 ? (raw + 15) / 16 : 1);
  if (MR > 48) return fail(ACNQP_ERR_INVALID, "site has too many rows for the tiled kernel (> 48 after padding)");
  std::vector<double> Gi((size_t)MR * NP, 0.0), lim(MR, 0.0);
  std::vector<int32_t> ty(MR, acnqp::kRowFree), abi(MR, -1);
  auto put = [&](int slot, int src_row, int type, double limit) {
    for (int i = 0; i < N; ++i) Gi[(size_t)slot * NP + i] = h->G[(size_t)src_row * N + i];
    abi[slot] = src_row;
    ty[slot] = type;
    lim[slot] = limit;
  };
  int peak_slot;   // first slot after the infrastructure rows: flat row (if any), then peak row
  if (h->cone == ACNQP_CONE_SOC) {
    for (int c = 0; c < M; ++c) {
      put(soc_slot(f64, c, 0), c, acnqp::kRowSocRe, h->limits[c]);
      put(soc_slot(f64, c, 1), c + M, acnqp::kRowSocIm, h->limits[c]);
    }
    peak_slot = f64 ? 8 * ((M + 3) / 4) : 2 * M;
  } else {
    for (int c = 0; c < M; ++c) put(c, c, acnqp::kRowBox, h->limits[c]);
    peak_slot = M;
  }
  int flat_slot = -1, max_slot = -1, pk_slot = -1;
  if (h->has_flat) { flat_slot = peak_slot; put(peak_slot, h->Mg - 1 - h->has_peak - h->has_max, acnqp::kRowQuad, 0.0); ++peak_slot; }
  if (h->has_max) { max_slot = peak_slot; put(peak_slot, h->Mg - 1 - h->has_peak, acnqp::kRowMax, 0.0); ++peak_slot; }
  if (h->has_peak) { pk_slot = peak_slot; put(peak_slot, h->Mg - 1, acnqp::kRowPeak, 0.0); }

  // Row equilibration (solver-internal, invisible at the ABI): every site row -- a SOC pair counts as
  // one row -- is scaled by 1/sqrt(|g|_2), and its limit with it.  The constraint set is unchanged;
  // the ADMM penalty now weighs a 26-EVSE feeder row and a 8-EVSE pod row alike, which roughly
  // halves the iteration count (measured on the Caltech-shaped network).
  std::vector<double> rs(MR, 1.0);
  for (int j = 0; j < MR; ++j) {
    if (ty[j] == acnqp::kRowFree || ty[j] == acnqp::kRowSocIm) continue;
    double n2 = 0;
    const int j2 = ty[j] == acnqp::kRowSocRe ? j + (f64 ? 4 : 1) : -1;
    for (int i = 0; i < N; ++i) {
      n2 += Gi[(size_t)j * NP + i] * Gi[(size_t)j * NP + i];
      if (j2 >= 0) n2 += Gi[(size_t)j2 * NP + i] * Gi[(size_t)j2 * NP + i];
    }
    const double sc = n2 > 0 ? 1.0 / std::sqrt(std::sqrt(n2)) : 1.0;
    rs[j] = sc;
    if (j2 >= 0) rs[j2] = sc;
  }
  for (int j = 0; j < MR; ++j) {
    for (int i = 0; i < N; ++i) Gi[(size_t)j * NP + i] *= rs[j];
    lim[j] *= rs[j];
  }
  d->flat_scale = flat_slot >= 0 ? rs[flat_slot] : 1.0;
  d->max_scale = max_slot >= 0 ? rs[max_slot] : 1.0;
  d->peak_scale = pk_slot >= 0 ? rs[pk_slot] : 1.0;

  std::vector<double> GGt((size_t)MR * MR, 0.0), lam, Q;
  for (int r = 0; r < MR; ++r)
    for (int c = 0; c < MR; ++c) {
      double s = 0;
      for (int i = 0; i < N; ++i) s += Gi[(size_t)r * NP + i] * Gi[(size_t)c * NP + i];
      GGt[(size_t)r * MR + c] = s;
    }
  jacobi_eigh(MR, GGt, lam, Q);
  double lmax = 0;
  for (int k = 0; k < MR; ++k) lmax = std::fmax(lmax, lam[k]);
  std::vector<double> Gh((size_t)MR * NP, 0.0);
  for (int k = 0; k < MR; ++k) {
    if (lam[k] < 1e-12 * std::fmax(1.0, lmax)) { lam[k] = 0.0; continue; }
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int r = 0; r < MR; ++r) s += Q[(size_t)r * MR + k] * Gi[(size_t)r * NP + i];
      Gh[(size_t)k * NP + i] = s;
    }
  }
  auto up = [&](void** dst, const std::vector<double>& src) -> hipError_t {
    std::vector<real> tmp(src.begin(), src.end());
    hipError_t e = hipMalloc(dst, tmp.size() * sizeof(real));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(real), hipMemcpyHostToDevice);
  };
  // MFMA A-operand fragments in the order the tiled and the large-site kernel read them (one coalesced
  // 64-lane row per fragment register): see acn_qp_tiled.hpp / acn_qp_stream.hpp.  NP / 16 EVSE tiles.
  std::vector<double> fragG, fragQ;
  {
    const int NWv = NP / 16, MT = MR / 16;
    fragG.assign((size_t)NWv * MT * 2 * 4 * 64, 0.0);
    fragQ.assign((size_t)MT * MT * 2 * 4 * 64, 0.0);
    for (int lane = 0; lane < 64; ++lane) {
      const int g = lane >> 4, t = lane & 15;
      for (int sI = 0; sI < 4; ++sI) {
        const int ro = acnqp::Mfma<real>::rowof(g, sI);
        for (int m = 0; m < MT; ++m) {
          for (int w = 0; w < NWv; ++w) {
            fragG[((((size_t)w * MT + m) * 2 + 0) * 4 + sI) * 64 + lane] = Gh[(size_t)(16 * m + t) * NP + 16 * w + ro];
            fragG[((((size_t)w * MT + m) * 2 + 1) * 4 + sI) * 64 + lane] = Gh[(size_t)(16 * m + ro) * NP + 16 * w + t];
          }
          for (int mi = 0; mi < MT; ++mi) {   // m plays the role of mo
            fragQ[((((size_t)m * MT + mi) * 2 + 0) * 4 + sI) * 64 + lane] = Q[(size_t)(16 * mi + ro) * MR + 16 * m + t];
            fragQ[((((size_t)m * MT + mi) * 2 + 1) * 4 + sI) * 64 + lane] = Q[(size_t)(16 * m + t) * MR + 16 * mi + ro];
          }
        }
      }
    }
  }
  hipError_t e = up(&d->G, Gi);
  if (e == hipSuccess && !fragG.empty()) e = up(&d->fragG, fragG);
  if (e == hipSuccess && !fragQ.empty()) e = up(&d->fragQ, fragQ);
  {
    // pair order: [k-slice / 2][lane][k-slice % 2] inside every 4 x 64 fragment block (one 16-byte load per lane
    // fetches two k-slices: acn_qp_long.hpp)
    auto paired = [](const std::vector<double>& in) {
      std::vector<double> out(in.size());
      for (size_t blk = 0; blk + 256 <= in.size(); blk += 256)
        for (int sI = 0; sI < 4; ++sI)
          for (int lane = 0; lane < 64; ++lane) out[blk + (size_t)(sI >> 1) * 128 + lane * 2 + (sI & 1)] = in[blk + (size_t)sI * 64 + lane];
      return out;
    };
    if (e == hipSuccess && !fragG.empty()) e = up(&d->fragG2, paired(fragG));
    if (e == hipSuccess && !fragQ.empty()) e = up(&d->fragQ2, paired(fragQ));
  }
  if (e == hipSuccess) e = up(&d->Ghat, Gh);
  if (e == hipSuccess) e = up(&d->Q, Q);
  if (e == hipSuccess) e = up(&d->lam, lam);
  if (e == hipSuccess) e = up(&d->rowlim, lim);
  if (e == hipSuccess) e = up(&d->rowscale, rs);
  if (e == hipSuccess) e = hipMalloc((void**)&d->rowtype, MR * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(d->rowtype, ty.data(), MR * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&d->rowabi, MR * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(d->rowabi, abi.data(), MR * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&d->Gabi, std::max<size_t>(h->G.size(), 1) * sizeof(double));
  if (e == hipSuccess && !h->G.empty()) e = hipMemcpy(d->Gabi, h->G.data(), h->G.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&d->limabi, std::max<size_t>(h->limits.size(), 1) * sizeof(double));
  if (e == hipSuccess && !h->limits.empty()) e = hipMemcpy(d->limabi, h->limits.data(), h->limits.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) { d->release(); return fail(ACNQP_ERR_HIP, std::string("site upload: ") + hipGetErrorString(e)); }
  d->MR = MR;
  d->ready = true;
  return ACNQP_OK;
}

// Shapes the register-resident tiled kernel takes: N <= 64, one column tile with any number of row tiles, or two
// column tiles with ONE row tile.  Two column tiles x two / three row tiles (horizon 17 ... 32 on a site of more than 16
// padded rows, e.g. the synthetic JPL site at horizon 24 of configs[2]) are not register-resident -- every wave would
// carry the whole site-row state redundantly (430-1,100 spilled registers; those instantiations are gone): two row
// tiles run through the LDS-resident variant of the long-horizon kernel (measured on 4,096 jpl52 x 24 problems: 189 ms
// tiled, 70 ms there), or its workspace variant with a demand-charge row / ACNQP_LDS_LONG=0 (diagnostic); three row
// tiles through the general-shape kernel.
static bool lds_long_shape(const acnqp_handle* h, int t_max) {
  const char* e = std::getenv("ACNQP_LDS_LONG");
  if (e && std::atoi(e) == 0) return false;
  return h->N <= 64 && t_max > 16 && t_max <= 32 && !h->has_max && h->dev64.MR == 32;
}
static bool tiled_shape(const acnqp_handle* h, int t_max, int k_sessions) {
  if (t_max > 16 && h->dev64.MR > 16) return false;   // two column tiles x two / three row tiles: not register-resident
  return h->N <= 64 && t_max <= 32 && k_sessions <= acnqp::kMaxK;
}

// shapes the large-site MFMA kernel takes (acn_qp_stream.hpp): wide sites, up to three column tiles
static bool stream_shape(const acnqp_handle* h, int t_max) {
  return h->N > 64 && t_max <= 48;
}

// shapes the long-horizon MFMA kernel takes (acn_qp_long.hpp): what the two kernels above leave, up to 288 periods
// and two row tiles, no demand-charge row
static bool long_shape(const acnqp_handle* h, int t_max, int k_sessions) {
  if (std::getenv("ACNQP_NO_LONG")) return false;   // diagnostic: the general-shape kernel instead
  return !tiled_shape(h, t_max, k_sessions) && !stream_shape(h, t_max) && t_max <= 288 && h->dev64.MR <= 32;
}

}  // namespace

extern "C" {

int32_t acnqp_abi_version(void) { return ACNQP_ABI_VERSION; }

const char* acnqp_last_error(void) { return g_last_error.c_str(); }

void acnqp_default_options(acnqp_options* o) {
  if (!o) return;
  o->eps_abs = 1e-8;
  o->eps_rel = 1e-8;
  o->max_iter = 20000;
  o->check_every = 20;
  o->adapt_every = 20;
  o->rho = 0.02;
  o->sigma = 1e-6;
  o->alpha = 1.4;
  o->adapt_tol = 3.0;
  o->reg_rel = 0.06;
  o->precision = 64;
  o->accel_mem = 5;
  o->stall_iters = 3000;
  o->retry_passes = 2;
  o->retry_max_iter = 8000;
  o->polish_iters = 800;
  o->retry_rho = 0.5;
  o->inaccurate_floor = 1e-5;
  o->polish_stall = 0;
}

int acnqp_create(const acnqp_site* site, int32_t device_id, acnqp_handle** out) {
  if (!site || !out) return fail(ACNQP_ERR_INVALID, "acnqp_create: null argument");
  *out = nullptr;
  const int N = site->n_evse, M = site->n_infra, Mg = site->n_rows;
  if (N < 1 || N > 1024)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: n_evse must be in [1, 1024]");
  if (site->cone != ACNQP_CONE_LINEAR && site->cone != ACNQP_CONE_SOC)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: cone must be ACNQP_CONE_LINEAR or ACNQP_CONE_SOC");
  const int expect = (site->cone == ACNQP_CONE_SOC ? 2 * M : M) + (site->has_peak ? 1 : 0) + (site->has_flat ? 1 : 0) + (site->has_max ? 1 : 0);
  if (M < 0 || Mg != expect || Mg > 48)
    return fail(ACNQP_ERR_INVALID, "acnqp_create: n_rows inconsistent with n_infra/cone/has_peak/has_flat/has_max or > 48");
  if (Mg > 0 && !site->G) return fail(ACNQP_ERR_INVALID, "acnqp_create: G is null");
  if (M > 0 && !site->limits) return fail(ACNQP_ERR_INVALID, "acnqp_create: limits is null");
  for (int j = 0; j < M; ++j)
    if (!(site->limits[j] >= 0)) return fail(ACNQP_ERR_INVALID, "acnqp_create: limits must be >= 0");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(ACNQP_ERR_NO_DEVICE, "acnqp_create: no HIP device visible");
  if (device_id < 0 || device_id >= ndev) return fail(ACNQP_ERR_INVALID, "acnqp_create: bad device_id");
  HIP_TRY(hipSetDevice(device_id));

  acnqp_handle* h = new acnqp_handle();
  h->device = device_id;
  h->N = N; h->M = M; h->Mg = Mg; h->cone = site->cone; h->has_peak = site->has_peak ? 1 : 0;
  h->has_flat = site->has_flat ? 1 : 0;
  h->has_max = site->has_max ? 1 : 0;
  h->NW = 4;
  h->NP = N <= 64 ? 64 : 16 * ((N + 15) / 16);
  if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || h->cus < 1) h->cus = 256;
  h->G.assign(site->G, site->G + (size_t)Mg * N);
  h->limits.assign(site->limits, site->limits + M);
  hipError_t e = hipSuccess;
  for (int k = 0; k < acnqp_handle::kEvRing && e == hipSuccess; ++k) {
    e = hipEventCreate(&h->ev_start[k]);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_stop[k]);
  }
  for (int k = 0; k < acnqp_handle::kSlots && e == hipSuccess; ++k) {
    e = hipStreamCreateWithFlags(&h->slot[k].st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->h2d_done[k], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    std::string msg = std::string("acnqp_create: ") + hipGetErrorString(e);
    acnqp_destroy(h);
    return fail(ACNQP_ERR_HIP, msg);
  }
  int rc = build_site_dev(h, &h->dev64);
  if (rc != ACNQP_OK) { acnqp_destroy(h); return rc; }
  if (hipMalloc((void**)&h->pol_stats, 16 * sizeof(int32_t)) != hipSuccess || hipMemset(h->pol_stats, 0, 16 * sizeof(int32_t)) != hipSuccess) {
    acnqp_destroy(h);
    return fail(ACNQP_ERR_HIP, "acnqp_create: polish counters");
  }
  *out = h;
  return ACNQP_OK;
}

void acnqp_destroy(acnqp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  h->dev64.release();
  for (int k = 0; k < acnqp_handle::kEvRing; ++k) {
    if (h->ev_start[k]) (void)hipEventDestroy(h->ev_start[k]);
    if (h->ev_stop[k]) (void)hipEventDestroy(h->ev_stop[k]);
  }
  for (auto& sl : h->slot) {
    if (sl.st) { (void)hipStreamSynchronize(sl.st); (void)hipStreamDestroy(sl.st); }
    sl.in.release();
    sl.out.release();
    sl.tin.release();
  }
  h->small_in.release();
  h->small_out.release();
  for (auto& ev : h->h2d_done) if (ev) (void)hipEventDestroy(ev);
  h->release_work();
  if (h->pol_stats) (void)hipFree(h->pol_stats);
  delete h;
}

// doubles of per-problem workspace the kernel that serves this shape needs (0: register / LDS resident)
static long long workspace_doubles(const acnqp_handle* h, int t_max, int k_sessions, int accel_req);

static int check_problem_shapes(const acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o,
                                const acnqp_results* r) {
  if (!h || !p || !o || !r) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null argument");
  if (p->batch < 0) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: negative batch");
  if (p->batch == 0) return ACNQP_OK;
  if (p->t_max < 1 || p->t_max > 4096)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: t_max must be in [1, 4096]");
  if (p->k_sessions < 1 || p->k_sessions > 4096)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: k_sessions must be in [1, 4096]");
  if (!p->horizon || !p->lb || !p->ub || !p->q || !p->pdiag || !p->s_off || !p->s_len || !p->s_cap || !p->s_eq)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null problem array");
  if (h->has_peak && !p->peak) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: site has a peak row but peak is null");
  if (h->has_flat && !p->lf) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: site has a flat row but lf is null");
  if (h->has_max && (!p->dc || !p->dfloor)) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: site has a max row but dc/dfloor is null");
  if ((p->warm_x == nullptr) != (p->warm_y == nullptr))
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: warm_x and warm_y must be given together");
  if (!r->x || !r->status || !r->iters || !r->pri_res || !r->dua_res || !r->obj)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: null result array");
  if (!(o->eps_abs >= 0) || !(o->eps_rel >= 0) || o->max_iter < 1 || o->check_every < 1 || !(o->rho > 0) ||
      !(o->sigma >= 0) || !(o->alpha > 0 && o->alpha < 2) || !(o->adapt_tol > 1) || !(o->reg_rel >= 0) ||
      o->adapt_every < 0 || o->polish_iters < 0 || o->polish_stall < 0 || o->stall_iters < 0 || o->retry_passes < 0 || o->retry_passes > 8 || o->retry_max_iter < 1 ||
      !(o->retry_rho > 0) || !(o->inaccurate_floor >= 0))
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: invalid option value");
  if (o->precision != 64)
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_batch: precision must be 64 (the fp32 loop was removed: it missed the "
                                   "1e-4 rate tolerance on weakly convex problems and was not faster; DESIGN.md section 3)");
  return ACNQP_OK;
}

int acnqp_solve_batch_device(acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o, acnqp_results* r,
                             void* hip_stream) {
  int rc = check_problem_shapes(h, p, o, r);
  if (rc != ACNQP_OK) return rc;
  if (p->batch == 0) return ACNQP_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  SiteDev* d = &h->dev64;
  acnqp::TiledArgs a;
  a.B = p->batch; a.N = h->N; a.Tm = p->t_max; a.K = p->k_sessions; a.NP = h->NP; a.MR = d->MR;
  a.G = d->G; a.Ghat = d->Ghat; a.Q = d->Q; a.lam = d->lam; a.rowlim = d->rowlim; a.rowtype = d->rowtype;
  a.fragG = d->fragG; a.fragQ = d->fragQ; a.fragG2 = d->fragG2; a.fragQ2 = d->fragQ2;
  a.horizon = p->horizon; a.lb = p->lb; a.ub = p->ub; a.q = p->q; a.pdiag = p->pdiag;
  a.s_off = p->s_off; a.s_len = p->s_len; a.s_cap = p->s_cap; a.s_eq = p->s_eq; a.peak = h->has_peak ? p->peak : nullptr;
  a.lf = h->has_flat ? p->lf : nullptr;
  a.dc = h->has_max ? p->dc : nullptr;
  a.dfloor = h->has_max ? p->dfloor : nullptr;
  a.warm_x = (p->warm_x && p->warm_y) ? p->warm_x : nullptr;
  a.warm_y = (p->warm_x && p->warm_y) ? p->warm_y : nullptr;
  a.y_out = r->y;
  a.rowabi = d->rowabi; a.rowscale = d->rowscale; a.Mg = h->Mg;
  a.x = r->x; a.status = r->status; a.iters = r->iters; a.pri = r->pri_res; a.dua = r->dua_res; a.obj = r->obj;
  a.eps_abs = o->eps_abs; a.eps_rel = o->eps_rel; a.rho0 = o->rho; a.sigma = o->sigma; a.alpha = o->alpha;
  a.adapt_tol = o->adapt_tol; a.reg_rel = o->reg_rel;
  a.max_iter = o->max_iter; a.check_every = o->check_every; a.adapt_every = o->adapt_every;
  a.peak_scale = d->peak_scale; a.flat_scale = d->flat_scale; a.max_scale = d->max_scale;
  a.accel_mem = std::max(0, o->accel_mem);
  a.stall_iters = o->stall_iters; a.retry_passes = o->retry_passes; a.retry_max_iter = o->retry_max_iter;
  a.retry_rho = o->retry_rho; a.inacc_floor = o->inaccurate_floor;
  a.pbuf_single = 0;
  a.order = nullptr;
  a.queue = nullptr;
  {
    // launches of the handle's own pipeline streams (the host-buffer entries: several launches in flight) keep their
    // workgroups only for a few problems each; a launch on a caller's stream is fully persistent
    bool own = false;
    for (auto& sl : h->slot) own = own || sl.st == st;
    a.grid_oversub = own ? 4 : 1;
  }
  a.polish_iters = 0; a.polish_stall = 0; a.resume = 0; a.y_for_polish_only = 0; a.pol_rows = 0; a.pol_list = nullptr; a.pol_count = nullptr; a.count_dev = nullptr;
  (void)hipGetLastError();   // drop any stale error so the checks below report this launch only
  // a problem whose workgroup never ran must not look solved (or carry the previous call's status)
  HIP_TRY(hipMemsetAsync(r->status, 0, (size_t)p->batch * sizeof(int32_t), st));
  // the launch's duration (acnqp_kernel_times, bench.py's roofline) covers everything the launch puts on the stream:
  // the two order kernels below as well as the solver kernel (they used to sit in front of the start event)
  const int evk = (int)(h->launches % acnqp_handle::kEvRing);
  HIP_TRY(hipEventRecord(h->ev_start[evk], st));
  static const bool no_order = std::getenv("ACNQP_NO_ORDER") != nullptr;   // diagnostic: queue position b = problem b
  static const bool no_queue_env = std::getenv("ACNQP_NO_QUEUE") != nullptr;   // diagnostic: the static schedule (workgroup w = position w)
  static const bool queue_all = std::getenv("ACNQP_QUEUE_ALL") != nullptr;     // diagnostic: the queue for the streaming kernels too
  // The work queue serves the kernels whose state is ON CHIP (register-resident, LDS-resident long-horizon): their
  // launches end on their slowest workgroup slot, and levelling the slots is worth 7-10 % (headline lone launch 28.8 ->
  // 26.9 ms, jpl52 x 24 x 4,096 67.2 -> 60.6 ms).  The kernels that stream their state through HBM are bandwidth-bound
  // across the whole chip: a slot that ends early leaves its bandwidth to the others, and the static schedule measured
  // the same or better (configs[4] leg 432 vs 434-438 ms, 54 x 144 x 2,048 237-243 vs 247-248 ms; gpurun_out/r4c):
  // they keep one workgroup per problem.
  // the wave-per-problem kernel's variant for this shape (0: another kernel family; acn_qp_wave.hip)
  const int wv = acnqp::wave_shape(h->N, p->t_max, p->k_sessions, d->MR, h->has_max, p->batch);
  const bool on_chip = wv > 0 || tiled_shape(h, p->t_max, p->k_sessions) || (long_shape(h, p->t_max, p->k_sessions) && lds_long_shape(h, p->t_max));
  const bool no_queue = no_queue_env || !(on_chip || queue_all);
  // (the order by sessions only for separable objectives: with a load-flattening or demand-charge row the coupling, not
  //  the number of sessions, sets the iteration count -- on the configs[4] leg it was 10 % SLOWER than the natural one)
  const bool want_order = p->batch >= kOrderMinBatch && !no_order && !h->has_flat && !h->has_max;
  acnqp_handle::Work* wk = h->work_for(st);
  if (!no_queue || want_order) {
    // scheduling buffer of this stream: [0] the queue counter (its own 256-byte line), then keys[B], order[B]
    const size_t need = 256 + (want_order ? (size_t)p->batch * 2 * sizeof(int32_t) : 0);
    if (need > wk->ord.cap) HIP_TRY(hipStreamSynchronize(st));   // an earlier launch on this stream may still read the old one
    HIP_TRY(wk->ord.reserve(need));
    if (!no_queue) {
      a.queue = static_cast<int32_t*>(wk->ord.p);
      HIP_TRY(hipMemsetAsync(a.queue, 0, sizeof(int32_t), st));
    }
    if (want_order) {
      int32_t* keys = static_cast<int32_t*>(wk->ord.p) + 64;
      int32_t* order = keys + p->batch;
      hipLaunchKernelGGL(order_keys_kernel, dim3((p->batch + 3) / 4), dim3(256), 0, st, p->s_len, p->k_sessions * h->N, p->batch, keys);
      hipLaunchKernelGGL(order_sort_kernel, dim3(1), dim3(kOrderKeys), 0, st, keys, p->batch, order);
      a.order = order;
      h->ordered_launches++;
    }
  }
  // workgroups of a work-queue launch that get a workspace of their own (the kernels that stream their state): at most
  // two resident workgroups per CU (no streaming kernel has more); the launchers cap their grid at it
  static const bool ws_per_problem = std::getenv("ACNQP_WS_PER_PROBLEM") != nullptr;   // diagnostic
  a.ws_by_slot = a.queue && !ws_per_problem ? 1 : 0;
  a.grid_cap = a.ws_by_slot ? std::min(p->batch, 2 * h->cus) : p->batch;
  const bool tiled = wv > 0 || tiled_shape(h, p->t_max, p->k_sessions);   // (no workspace: state on chip)
  const bool stream = !tiled && stream_shape(h, p->t_max);
  const bool lng = !tiled && long_shape(h, p->t_max, p->k_sessions);
  acnqp::GeneralArgs ga;
  acnqp::StreamArgs sa;
  if (stream) {
    // large-site kernel: iterates streamed through a per-problem workspace in MFMA fragment order
    const int CT = (p->t_max + 15) / 16;
    a.accel_mem = std::min(a.accel_mem, acnqp::kStreamAccelMax);
    sa.ws_per_problem = acnqp::stream_workspace(h->NP, CT, p->k_sessions, d->MR / 16, a.accel_mem);
    DevBuf* wsb = &wk->buf;
    const size_t need = (size_t)sa.ws_per_problem * a.grid_cap * sizeof(double);
    if (need > wsb->cap) HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(wsb->reserve(need));
    sa.work = static_cast<double*>(wsb->p);
    sa.t = a;
  } else if (lng) {
    // long-horizon kernel: same workspace idea, one more array (r0 / zh)
    a.accel_mem = std::min(a.accel_mem, acnqp::kLongAccelMax);
    sa.ws_per_problem = acnqp::long_workspace(h->NP, acnqp::long_tiles(p->t_max), p->k_sessions, d->MR / 16, a.accel_mem);
    DevBuf* wsb = &wk->buf;
    const size_t need = (size_t)sa.ws_per_problem * a.grid_cap * sizeof(double);
    if (need > wsb->cap) HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(wsb->reserve(need));
    sa.work = static_cast<double*>(wsb->p);
    sa.t = a;
  } else if (!tiled) {
    // general-shape fallback: state streamed through a global workspace (not graph-capturable: it may allocate)
    const size_t rsz = 8;
    ga.ws_per_problem = workspace_doubles(h, p->t_max, p->k_sessions, a.accel_mem);
    DevBuf* wsb = &wk->buf;
    const size_t need = (size_t)ga.ws_per_problem * a.grid_cap * rsz;
    if (need > wsb->cap) HIP_TRY(hipStreamSynchronize(st));   // an earlier launch on this stream may still use the old buffer
    HIP_TRY(wsb->reserve(need));
    ga.work = wsb->p;
    ga.pair_stride = 4;
    ga.t = a;
  }
  auto launch_solver = [&](const acnqp::TiledArgs& aa) -> hipError_t {
    if (wv > 0) return acnqp::launch_wave(aa, st);
    if (tiled) return p->t_max <= 16 ? acnqp::launch_tiled_ct1(aa, st) : acnqp::launch_tiled_ct2(aa, st);
    if (stream) { sa.t = aa; return acnqp::launch_stream(sa, st); }
    if (lng) { sa.t = aa; return acnqp::launch_long(sa, st, lds_long_shape(h, p->t_max)); }
    // workgroup size by problem size: the plain loops are latency-bound, more threads per problem hide more of it
    const long long nvar = (long long)h->N * p->t_max;
    const int nt = nvar <= 4096 ? 256 : (nvar <= 12288 ? 512 : 1024);
    ga.t = aa;
    return acnqp::launch_general(ga, nt, st);
  };
  // ---- the polish (acn_qp_polish.hpp): small sites, separable objective, served by an on-chip kernel ------------------
  // solver kernel (pass 0 up to polish_iters iterations; what has not converged by then is listed) -> polish kernel over
  // the list -> solver kernel again over the list for what the polish gave up on (from scratch, with the retry passes:
  // the answer it had before there was a polish).  Three launches on the stream, the last two nearly empty as a rule.
  const int nrow_site = h->M + h->has_peak;
  int pol_blk = -1;   // LDS doubles for the blocks of the polish's system; -1: no polish for this launch
  static const bool no_polish = std::getenv("ACNQP_NO_POLISH") != nullptr;   // diagnostic
  if (!no_polish && o->polish_iters > 0 && o->polish_iters < o->max_iter && on_chip && h->N <= 64 && p->t_max <= 32 &&
      p->k_sessions <= acnqp::kMaxK && !h->has_flat && !h->has_max && !a.warm_x && nrow_site > 0) {
    // LDS of a polish workgroup: where the solver kernel runs two workgroups per CU (the headline shape: one column tile,
    // one row tile, one session slot: 77 KB each) the polish takes no more than one of those slots, so that it starts as
    // soon as ANY solver workgroup of a neighbouring stream's launch ends; elsewhere the whole CU
    const bool two_per_cu = tiled && p->t_max <= 16 && d->MR == 16 && p->k_sessions == 1;
    pol_blk = acnqp::polish_blocks_that_fit(h->N, p->t_max, h->Mg, nrow_site, acnqp::polish_max_sess(h->N, p->k_sessions),
                                            two_per_cu ? 76 * 1024 : 160 * 1024);
    if (pol_blk < 2 * nrow_site * (2 * nrow_site + 1) / 2) pol_blk = -1;   // not even one full block
    // The polish is a remedy for a launch's TAIL, and its phase runs behind the solver launch.  A launch of many problems
    // per resident slot of a one-workgroup-per-CU shape is throughput-bound -- its stragglers overlap the bulk of the work
    // -- so the polish phase (horizon-24 rounds: 0.1 ms each, up to 30 of them) only adds to it: jpl52 x 24 x 4,096
    // 61.0 -> 67.8 ms with it.  Such launches keep the ADMM alone (ACNQP_POLISH_ALWAYS=1: diagnostic).
    static const bool polish_always = std::getenv("ACNQP_POLISH_ALWAYS") != nullptr;
    if (!two_per_cu && p->batch > 8 * h->cus && !polish_always) pol_blk = -1;
  }
  hipError_t e = hipSuccess;
  if (pol_blk >= 0) {
    // [0] queue of the polish kernel, [1] queue of the resume launch, [2] list length; list[B]; multipliers [B][Mg][Tm]
    // unless the caller wants them anyway; then the polish's global scratch for Schur systems beyond its LDS
    const int pol_max = acnqp::polish_max_rows(nrow_site, p->t_max);
    const int pol_grid = std::max(1, std::min(p->batch, a.grid_oversub > 1 ? 32 : h->cus));
    const size_t ybytes = r->y ? 0 : (size_t)p->batch * h->Mg * p->t_max * sizeof(double);
    const size_t lbytes = ((size_t)p->batch * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t need = 256 + lbytes + ybytes;
    if (need > wk->pol.cap) HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(wk->pol.reserve(need));
    int32_t* ctr = static_cast<int32_t*>(wk->pol.p);
    int32_t* list = ctr + 64;
    double* ybuf = r->y ? r->y : reinterpret_cast<double*>(static_cast<char*>(wk->pol.p) + 256 + lbytes);
    HIP_TRY(hipMemsetAsync(ctr, 0, 256, st));
    acnqp::TiledArgs a1 = a;
    a1.polish_iters = o->polish_iters - o->polish_iters % std::max(1, o->check_every);   // the exit is taken at a residual check
    if (a1.polish_iters < o->check_every) a1.polish_iters = o->check_every;
    a1.pol_list = list; a1.pol_count = ctr + 2; a1.y_out = ybuf; a1.pol_rows = pol_max;
    // (measured, wave kernel: the early hand-over takes 5 % off the headline launch and 23 % off configs[3] site 0, and hands
    //  ten times as many problems to the polish -- one 256-batch per call 81 -> 62 k QP/s, the table path 654 -> 580 k: off
    //  by default; options.polish_stall, ABI v9, is the caller's switch, the environment variable the diagnostic one)
    static const int early = std::getenv("ACNQP_EARLY_HANDOVER") ? std::atoi(std::getenv("ACNQP_EARLY_HANDOVER")) : 0;   // diagnostic: the window (1: polish_iters / 4)
    a1.polish_stall = early > 1 ? early : (early == 1 ? std::max(o->check_every, a1.polish_iters / 4) : o->polish_stall);
    a1.y_for_polish_only = r->y ? 0 : 1;
    e = launch_solver(a1);
    if (e == hipSuccess) {
      acnqp::PolishArgs pa;
      pa.B = p->batch; pa.N = h->N; pa.Tm = p->t_max; pa.K = p->k_sessions; pa.M = h->M; pa.Mg = h->Mg; pa.cone = h->cone;
      pa.has_peak = h->has_peak; pa.max_rows = pol_max; pa.blk_doubles = pol_blk; pa.max_sess = acnqp::polish_max_sess(h->N, p->k_sessions);
      pa.G = d->Gabi; pa.limits = d->limabi;
      pa.horizon = p->horizon; pa.lb = p->lb; pa.ub = p->ub; pa.q = p->q; pa.pdiag = p->pdiag;
      pa.s_off = p->s_off; pa.s_len = p->s_len; pa.s_cap = p->s_cap; pa.s_eq = p->s_eq; pa.peak = h->has_peak ? p->peak : nullptr;
      pa.x = r->x; pa.y = ybuf; pa.status = r->status; pa.iters = r->iters; pa.pri = r->pri_res; pa.dua = r->dua_res; pa.obj = r->obj;
      pa.list = list; pa.count = ctr + 2; pa.queue = ctr + 0; pa.stats = h->pol_stats; pa.reg_rel = o->reg_rel;
      // (pipelined chunks hand a handful of problems over: few workgroups, so that the launch does not queue for 256 slots
      //  behind the neighbouring streams' solver launches)
      e = acnqp::launch_polish(pa, pol_grid, st);
    }
    if (e == hipSuccess) {
      acnqp::TiledArgs a3 = a;
      a3.resume = 1; a3.order = list; a3.count_dev = ctr + 2; a3.queue = ctr + 1;
      // ... from where the first launch left them: its schedule and site-row multipliers are the warm start of pass 0 (the
      // retry passes that may follow start cold, as always).  Measured on jpl52 x 24 x 4,096, where two problems have more
      // tight rows than the polish holds: from scratch the launch took 93 ms against 61 ms without a polish.
      a3.warm_x = r->x; a3.warm_y = ybuf;
      a3.ws_by_slot = 1; a3.grid_cap = std::min(p->batch, a.grid_oversub > 1 ? 64 : 2 * h->cus);   // (<= the grid the workspace was sized for)
      if (!a.ws_by_slot) a3.grid_cap = std::min(a3.grid_cap, a.grid_cap);
      e = launch_solver(a3);
    }
  } else {
    e = launch_solver(a);
  }
  if (e != hipSuccess) return fail(ACNQP_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  HIP_TRY(hipEventRecord(h->ev_stop[evk], st));
  if (wk->last) HIP_TRY(hipEventRecord(wk->last, st));   // what an eviction of this stream's buffers waits for
  ++h->launches;
  return ACNQP_OK;
}

static long long workspace_doubles(const acnqp_handle* h, int t_max, int k_sessions, int accel_req) {
  const SiteDev* d = &h->dev64;
  if (tiled_shape(h, t_max, k_sessions)) return 0;
  if (stream_shape(h, t_max))
    return acnqp::stream_workspace(h->NP, (t_max + 15) / 16, k_sessions, d->MR / 16, std::min(std::max(0, accel_req), acnqp::kStreamAccelMax));
  if (long_shape(h, t_max, k_sessions))
    return acnqp::long_workspace(h->NP, acnqp::long_tiles(t_max), k_sessions, d->MR / 16, std::min(std::max(0, accel_req), acnqp::kLongAccelMax));
  const long long n = (long long)h->N * t_max, mt = (long long)d->MR * t_max, Dn = n + mt;
  const int gm = std::min(std::max(0, accel_req), acnqp::kGenAccelMax);
  // solver state, the certificate's dual snapshot, the Anderson vectors (u, f: reals; correction and rings: floats)
  return 7 * n + 8 * mt + 3LL * k_sessions * h->N + 8 + 2 * Dn + ((1 + 2LL * gm) * Dn * 4 + 7) / 8 + 2;
}

int32_t acnqp_accel_columns(acnqp_handle* h, int32_t t_max, int32_t k_sessions, int32_t precision, int32_t requested) {
  if (!h || t_max < 1 || k_sessions < 1 || requested <= 0) return 0;
  if (precision != 64) return 0;
  if (acnqp::wave_shape(h->N, t_max, k_sessions, h->dev64.MR, h->has_max, 1 << 30) > 0) return std::min(requested, acnqp::wave_accel_columns());
  if (!tiled_shape(h, t_max, k_sessions)) {
    if (stream_shape(h, t_max)) return std::min(requested, acnqp::kStreamAccelMax);   // ring in its workspace
    if (long_shape(h, t_max, k_sessions)) return std::min(requested, acnqp::kLongAccelMax);   // ring in its workspace
    return std::min(requested, acnqp::kGenAccelMax);     // general-shape kernel: ring in its workspace
  }
  SiteDev* d = &h->dev64;
  const int CT = (t_max + 15) / 16, MT = d->MR / 16;
  int single = 0;
  return std::min(requested, acnqp::accel_capacity_best(4, MT, CT, h->NP, k_sessions, &single));
}

static float event_pair_ms(acnqp_handle* h, long long launch) {
  const int k = (int)(launch % acnqp_handle::kEvRing);
  if (hipEventSynchronize(h->ev_stop[k]) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, h->ev_start[k], h->ev_stop[k]) != hipSuccess) return -1.0f;
  return ms;
}

int64_t acnqp_launch_count(acnqp_handle* h) { return h ? (int64_t)h->launches : 0; }
int64_t acnqp_ordered_launch_count(acnqp_handle* h) { return h ? (int64_t)h->ordered_launches : 0; }

int acnqp_polish_stats(acnqp_handle* h, int64_t* out, int32_t capacity) {
  if (!h || !out || capacity < 1) return fail(ACNQP_ERR_INVALID, "acnqp_polish_stats: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  int32_t v[16] = {0};
  HIP_TRY(hipMemcpy(v, h->pol_stats, sizeof(v), hipMemcpyDeviceToHost));
  for (int k = 0; k < capacity && k < 16; ++k) out[k] = v[k];
  return ACNQP_OK;
}

#ifdef ACNQP_DEBUG_WS
// diagnostic builds only (tools/gpu_long_race.py): the most recently used kernel workspace, copied to the host
extern "C" int64_t acnqp_debug_copy_workspace(acnqp_handle* h, double* out, int64_t n_doubles) {
  if (!h || h->work.empty()) return -1;
  size_t best = 0;
  for (size_t k = 1; k < h->work.size(); ++k) if (h->work[k].used > h->work[best].used) best = k;
  (void)hipDeviceSynchronize();
  const int64_t have = (int64_t)(h->work[best].buf.cap / sizeof(double));
  const int64_t n = std::min(have, n_doubles);
  if (out && n > 0 && hipMemcpy(out, h->work[best].buf.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return have;
}
#endif

float acnqp_last_kernel_ms(acnqp_handle* h) {
  if (!h || h->launches == 0) return -1.0f;
  return event_pair_ms(h, h->launches - 1);
}

int32_t acnqp_kernel_times(acnqp_handle* h, float* out_ms, int32_t capacity) {
  if (!h || !out_ms || capacity <= 0) return 0;
  long long first = std::max(h->reported, h->launches - acnqp_handle::kEvRing);
  first = std::max(first, h->launches - (long long)capacity);
  int32_t n = 0;
  for (long long l = first; l < h->launches; ++l) out_ms[n++] = event_pair_ms(h, l);
  h->reported = h->launches;
  return n;
}

// ---- host-buffer entry points ------------------------------------------------------------------------
namespace {

struct Piece { int g; long long lo, n, pos; };   // problems [lo, lo + n) of batch g sit at [pos, pos + n) of their chunk

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// device layout of one chunk of `cn` problems (inputs in slot.in, results in slot.out)
struct ChunkLayout {
  size_t lb, ub, q, pd, hz, so, sl, sc, eq, pk, lf, dc, df, wx, wy, in_total;
  size_t x, st, it, pr, du, ob, y, out_total;
  ChunkLayout(size_t cn, size_t N, size_t Tm, size_t K, size_t Mg, bool peak, bool flat, bool mx, bool warm, bool want_y) {
    size_t o = 0;
    lb = o; o += al256(cn * N * Tm * 8);
    ub = o; o += al256(cn * N * Tm * 8);
    q = o;  o += al256(cn * N * Tm * 8);
    pd = o; o += al256(cn * 8);
    hz = o; o += al256(cn * 4);
    so = o; o += al256(cn * K * N * 4);
    sl = o; o += al256(cn * K * N * 4);
    sc = o; o += al256(cn * K * N * 8);
    eq = o; o += al256(cn);
    pk = o; o += al256(peak ? cn * Tm * 8 : 0);
    lf = o; o += al256(flat ? cn * 8 : 0);
    dc = o; o += al256(mx ? cn * 8 : 0);
    df = o; o += al256(mx ? cn * 8 : 0);
    wx = o; o += al256(warm ? cn * N * Tm * 8 : 0);
    wy = o; o += al256(warm ? cn * Mg * Tm * 8 : 0);
    in_total = o;
    o = 0;
    x = o;  o += al256(cn * N * Tm * 8);
    st = o; o += al256(cn * 4);
    it = o; o += al256(cn * 4);
    pr = o; o += al256(cn * 8);
    du = o; o += al256(cn * 8);
    ob = o; o += al256(cn * 8);
    y = o;  o += al256(want_y ? cn * Mg * Tm * 8 : 0);
    out_total = o;
  }
};

// per_problem_bytes: inputs + results staged per problem -- every one of the kSlots pipeline slots holds a chunk of
// each, so a chunk is capped at 1 GiB of the sum
long long chunk_problems(size_t per_problem_bytes, bool on_chip, bool wave = false) {
  // problems per launch: large enough that the launch tail (its slowest problems) is short.  The register-resident
  // kernel takes 1,024 (two per workgroup slot): with the launches sorted longest-first (order_sort_kernel) and four
  // streams in flight the shorter pipeline head and tail outweigh the per-launch tails (bench: 441 -> 455 k QP/s)
  long long want = on_chip ? 1024 : 2048;
  // the wave-per-problem kernel's launches are persistent (four problems in flight per CU, one wave each): a launch ends
  // with its slowest problem whatever its size, so few large chunks (sweep at 16,384 problems per call: 4,096 / 6,144 /
  // 7,168 / 8,192 / 9,216 -> 18.9 / 19.0 / 18.3 / 18.0 / 19.4 ms: what matters beside the size is that the LAST chunk is
  // a small one -- plan_chunks below)
  if (wave) want = 8192;
  if (const char* e = std::getenv("ACNQP_CHUNK")) { const long long v = std::atoll(e); if (v > 0) want = v; }
  const long long by_mem = (long long)((size_t)1024 * 1024 * 1024 / std::max<size_t>(per_problem_bytes, 1));
  long long n = std::max<long long>(1, std::min(want, by_mem));
  if (n >= 512) n -= n % 512;   // whole rounds of the chip's 512 workgroup slots (2 per CU): no thin last round
  return n;
}

// what the caller's small result arrays still need after the streams have drained: a copy out of the pinned mirror
struct Scatter { int g; size_t lo, n, pos; const char* host; size_t st, it, pr, du, ob; };

// Chunk sizes of a call of `total` problems of ONE shape served by the wave kernel: a quarter-size and a half-size chunk
// in front (the first kernel waits for its inputs, and nothing overlaps that copy), full chunks, and a quarter-size one
// at the END (what follows the last solver launch -- its polish, its result copies -- is exposed too):
// c/4 + c/2 + k c + c/4 = total with the smallest k for which c <= cap.
void plan_chunks(long long total, long long cap, std::vector<long long>* out) {
  std::vector<long long>& plan = *out;
  plan.clear();
  if (const char* e = std::getenv("ACNQP_PLAN")) {   // diagnostic: explicit chunk sizes "a,b,c" (the rest goes into a last chunk)
    for (const char* q = e; *q;) { plan.push_back(std::atoll(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    plan.push_back(total);
    return;
  }
  if (total < 2048 || cap < 2048) { plan.push_back(total); return; }
  const long long k = (total + cap - 1) / cap - 1;
  long long c = total / (k + 1);
  c = std::max<long long>(2048, std::min(cap, c - c % 512));
  plan.push_back(c / 4);
  plan.push_back(c / 2);
  for (long long i = 0; i < k; ++i) plan.push_back(c);
  plan.push_back(total);   // (the rest: the loop below ends a chunk when the call's problems are gone)
}

int run_pipeline(acnqp_handle* h, int nb, const acnqp_problems* P, const acnqp_options* o, acnqp_results* R, std::vector<Scatter>* scatter) {
  const size_t N = h->N;
  const bool peak = h->has_peak, flat = h->has_flat, mx = h->has_max;
  long long call_total = 0;
  bool uniform_call = true;   // one shape, one set of optional arrays: the call's chunks can be planned as a whole
  std::vector<long long> plan;
  for (int g = 0; g < nb; ++g) {
    call_total += P[g].batch;
    uniform_call = uniform_call && P[g].t_max == P[0].t_max && P[g].k_sessions == P[0].k_sessions &&
                   (P[g].warm_x != nullptr) == (P[0].warm_x != nullptr) && (R[g].y != nullptr) == (R[0].y != nullptr);
  }
  // chunks: consecutive batches of one shape (t_max, k_sessions) share launches of up to chunk_problems() problems
  std::vector<std::vector<Piece>> chunks;
  long long fill = 0, cap = 0;
  int cur_T = -1, cur_K = -1, cur_opt = -1;
  for (int g = 0; g < nb; ++g) {
    const long long B = P[g].batch;
    if (B == 0) continue;
    const size_t Tm = P[g].t_max, K = P[g].k_sessions;
    const int opt = (P[g].warm_x ? 1 : 0) | (R[g].y ? 2 : 0);   // chunks are uniform in warm start / multiplier output
    for (long long lo = 0; lo < B;) {
      if (chunks.empty() || (int)Tm != cur_T || (int)K != cur_K || opt != cur_opt || fill >= cap) {
        chunks.emplace_back();
        cur_T = (int)Tm; cur_K = (int)K; cur_opt = opt; fill = 0;
        // (the kernels' workspaces belong to the resident workgroup slots since the work queue: no per-problem term)
        cap = chunk_problems(4 * N * Tm * 8 + K * N * 16 + Tm * 8 + 96, tiled_shape(h, (int)Tm, (int)K),
                             acnqp::wave_shape(h->N, (int)Tm, (int)K, h->dev64.MR, h->has_max, (int)std::min<long long>(call_total, 1 << 30)) > 0);
        // ramp: the first kernel cannot start before its chunk's H2D has landed, and nothing overlaps that copy -- a
        // quarter-size first chunk (then a half-size one) shortens the exposed head of the pipeline
        static const bool ramp = std::getenv("ACNQP_NO_RAMP") == nullptr;
        const bool is_wave = acnqp::wave_shape(h->N, (int)Tm, (int)K, h->dev64.MR, h->has_max, (int)std::min<long long>(call_total, 1 << 30)) > 0;
        if (ramp && is_wave && uniform_call) {
          if (plan.empty()) plan_chunks(call_total, cap, &plan);
          cap = chunks.size() <= plan.size() ? plan[chunks.size() - 1] : cap;
        } else if (ramp && cap >= 1024) {
          if (chunks.size() == 1) cap /= 4;
          else if (chunks.size() == 2) cap /= 2;
        }
      }
      const long long n = std::min(B - lo, cap - fill);
      chunks.back().push_back({g, lo, n, fill});
      fill += n; lo += n;
    }
  }
  // pinned mirrors of the chunks' small arrays (device ranges [pd, wx) and [st, y) of ChunkLayout), whole call; beyond
  // kSmallCap the per-batch copies of old (a call that large is not bound by their latency)
  const size_t Mg0 = (size_t)h->Mg;
  constexpr size_t kSmallCap = (size_t)256 << 20;
  std::vector<size_t> in_off(chunks.size()), out_off(chunks.size());
  size_t in_sum = 0, out_sum = 0;
  for (size_t c = 0; c < chunks.size(); ++c) {
    const std::vector<Piece>& pcs = chunks[c];
    const size_t cn = (size_t)(pcs.back().pos + pcs.back().n);
    const ChunkLayout L(cn, N, P[pcs[0].g].t_max, P[pcs[0].g].k_sessions, Mg0, peak, flat, mx, P[pcs[0].g].warm_x != nullptr, R[pcs[0].g].y != nullptr);
    in_off[c] = in_sum; in_sum += L.wx - L.pd;
    out_off[c] = out_sum; out_sum += L.y - L.st;
  }
  static const bool no_stage = std::getenv("ACNQP_NO_STAGING") != nullptr;   // diagnostic: the per-batch copies
  const bool staged = !no_stage && scatter != nullptr && in_sum + out_sum <= kSmallCap;
  if (staged) {
    HIP_TRY(h->small_in.reserve(in_sum));
    HIP_TRY(h->small_out.reserve(out_sum));
  }
  for (size_t c = 0; c < chunks.size(); ++c) {
    acnqp_handle::Slot& S = h->slot[c % acnqp_handle::kSlots];
    const std::vector<Piece>& pcs = chunks[c];
    const size_t cn = (size_t)(pcs.back().pos + pcs.back().n);
    const size_t Tm = P[pcs[0].g].t_max, K = P[pcs[0].g].k_sessions;
    const bool warm = P[pcs[0].g].warm_x != nullptr, want_y = R[pcs[0].g].y != nullptr;
    const size_t Mg = (size_t)h->Mg;
    const ChunkLayout L(cn, N, Tm, K, Mg, peak, flat, mx, warm, want_y);
    char* hs = staged ? static_cast<char*>(h->small_in.p) + in_off[c] - L.pd : nullptr;   // hs + L.field = the mirror of di + L.field
    if (L.in_total > S.in.cap || L.out_total > S.out.cap) HIP_TRY(hipStreamSynchronize(S.st));   // staging still in use
    HIP_TRY(S.in.reserve(L.in_total));
    HIP_TRY(S.out.reserve(L.out_total));
    char* di = static_cast<char*>(S.in.p);
    char* dq = static_cast<char*>(S.out.p);
    // the chunks' input copies one chunk after the other (they share the link anyway): the first kernel's inputs do not
    // wait for a share of the bandwidth the second chunk's copies would take
    static const bool chain = std::getenv("ACNQP_NO_H2D_CHAIN") == nullptr;
    if (chain && c > 0) HIP_TRY(hipStreamWaitEvent(S.st, h->h2d_done[(c - 1) % acnqp_handle::kSlots], 0));
    for (const Piece& pc : pcs) {
      const acnqp_problems& p = P[pc.g];
      const size_t lo = (size_t)pc.lo, n = (size_t)pc.n, pos = (size_t)pc.pos;
      const size_t nv = N * Tm, ns = K * N;
#define H2D(field, base, elem, per)                                                                             \
  HIP_TRY(hipMemcpyAsync(di + (base) + pos * (per) * (elem), reinterpret_cast<const char*>(p.field) + lo * (per) * (elem), \
                         n * (per) * (elem), hipMemcpyHostToDevice, S.st))
// small arrays: into the pinned mirror (one H2D per chunk below); without staging, a copy each
#define H2S(field, base, elem, per)                                                                             \
  do {                                                                                                          \
    if (staged) std::memcpy(hs + (base) + pos * (per) * (elem), reinterpret_cast<const char*>(p.field) + lo * (per) * (elem), n * (per) * (elem)); \
    else H2D(field, base, elem, per);                                                                           \
  } while (0)
      H2D(lb, L.lb, 8, nv);
      H2D(ub, L.ub, 8, nv);
      H2D(q, L.q, 8, nv);
      H2S(pdiag, L.pd, 8, 1);
      H2S(horizon, L.hz, 4, 1);
      H2S(s_off, L.so, 4, ns);
      H2S(s_len, L.sl, 4, ns);
      H2S(s_cap, L.sc, 8, ns);
      H2S(s_eq, L.eq, 1, 1);
      if (peak) H2S(peak, L.pk, 8, Tm);
      if (flat) H2S(lf, L.lf, 8, 1);
      if (mx) { H2S(dc, L.dc, 8, 1); H2S(dfloor, L.df, 8, 1); }
      if (warm) { H2D(warm_x, L.wx, 8, nv); H2D(warm_y, L.wy, 8, Mg * Tm); }
#undef H2S
#undef H2D
    }
    if (staged) HIP_TRY(hipMemcpyAsync(di + L.pd, hs + L.pd, L.wx - L.pd, hipMemcpyHostToDevice, S.st));
    if (chain) HIP_TRY(hipEventRecord(h->h2d_done[c % acnqp_handle::kSlots], S.st));
    acnqp_problems dp;
    dp.batch = (int32_t)cn; dp.t_max = (int32_t)Tm; dp.k_sessions = (int32_t)K;
    dp.lb = reinterpret_cast<const double*>(di + L.lb);
    dp.ub = reinterpret_cast<const double*>(di + L.ub);
    dp.q = reinterpret_cast<const double*>(di + L.q);
    dp.pdiag = reinterpret_cast<const double*>(di + L.pd);
    dp.horizon = reinterpret_cast<const int32_t*>(di + L.hz);
    dp.s_off = reinterpret_cast<const int32_t*>(di + L.so);
    dp.s_len = reinterpret_cast<const int32_t*>(di + L.sl);
    dp.s_cap = reinterpret_cast<const double*>(di + L.sc);
    dp.s_eq = reinterpret_cast<const uint8_t*>(di + L.eq);
    dp.peak = peak ? reinterpret_cast<const double*>(di + L.pk) : nullptr;
    dp.lf = flat ? reinterpret_cast<const double*>(di + L.lf) : nullptr;
    dp.dc = mx ? reinterpret_cast<const double*>(di + L.dc) : nullptr;
    dp.dfloor = mx ? reinterpret_cast<const double*>(di + L.df) : nullptr;
    dp.warm_x = warm ? reinterpret_cast<const double*>(di + L.wx) : nullptr;
    dp.warm_y = warm ? reinterpret_cast<const double*>(di + L.wy) : nullptr;
    acnqp_results dr;
    dr.x = reinterpret_cast<double*>(dq + L.x);
    dr.status = reinterpret_cast<int32_t*>(dq + L.st);
    dr.iters = reinterpret_cast<int32_t*>(dq + L.it);
    dr.pri_res = reinterpret_cast<double*>(dq + L.pr);
    dr.dua_res = reinterpret_cast<double*>(dq + L.du);
    dr.obj = reinterpret_cast<double*>(dq + L.ob);
    dr.x_dev = nullptr;
    dr.y = want_y ? reinterpret_cast<double*>(dq + L.y) : nullptr;
    const int rc = acnqp_solve_batch_device(h, &dp, o, &dr, S.st);
    if (rc != ACNQP_OK) return rc;
    char* ho = staged ? static_cast<char*>(h->small_out.p) + out_off[c] - L.st : nullptr;   // ho + L.field = the mirror of dq + L.field
    if (staged) HIP_TRY(hipMemcpyAsync(ho + L.st, dq + L.st, L.y - L.st, hipMemcpyDeviceToHost, S.st));
    for (const Piece& pc : pcs) {
      const acnqp_results& r = R[pc.g];
      const size_t lo = (size_t)pc.lo, n = (size_t)pc.n, pos = (size_t)pc.pos;
#define D2H(field, base, elem, per)                                                                          \
  HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(r.field) + lo * (per) * (elem), dq + (base) + pos * (per) * (elem), \
                         n * (per) * (elem), hipMemcpyDeviceToHost, S.st))
      D2H(x, L.x, 8, N * Tm);
      if (staged) {
        scatter->push_back(Scatter{pc.g, lo, n, pos, ho, L.st, L.it, L.pr, L.du, L.ob});
      } else {
        D2H(status, L.st, 4, 1);
        D2H(iters, L.it, 4, 1);
        D2H(pri_res, L.pr, 8, 1);
        D2H(dua_res, L.du, 8, 1);
        D2H(obj, L.ob, 8, 1);
      }
      if (want_y) D2H(y, L.y, 8, Mg * Tm);
#undef D2H
      if (r.x_dev)
        HIP_TRY(hipMemcpyAsync(r.x_dev + lo * N * Tm, dq + L.x + pos * N * Tm * 8, n * N * Tm * 8, hipMemcpyDeviceToDevice, S.st));
    }
  }
  return ACNQP_OK;
}

}  // namespace

// ---- session-table entry: the dense problem arrays are formed on the device ---------------------------------------------
namespace {

struct TableExpandArgs {
  int N, Tm, K;
  long long s_base, r_base;          // first session / rate entry of the chunk (the segment arrays hold global indices)
  const int32_t *q_index, *sess_seg, *s_evse, *s_slot, *s_off, *s_len, *rate_seg;
  const double *q_table, *s_cap, *min_rates, *max_rates;
  double *lb, *ub, *q, *sc;
  int32_t *so, *sl;
};

// One workgroup per problem of the chunk: zero its bounds and session slots, copy its horizon's linear cost, then
// scatter its sessions -- lb / ub over the window (aco.py:62-75, ub < lb -> lb) and (offset, length, cap) into the
// EVSE's slot (aco.py:105-123).  Windows of one EVSE are disjoint: no two sessions write one entry.
__global__ __launch_bounds__(256) void table_expand_kernel(const TableExpandArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t nv = (size_t)a.N * a.Tm, ns = (size_t)a.K * a.N;
  double *lb = a.lb + b * nv, *ub = a.ub + b * nv, *q = a.q + b * nv;
  const double* qt = a.q_table + (size_t)a.q_index[b] * nv;
  for (size_t k = tid; k < nv; k += 256) { lb[k] = 0.0; ub[k] = 0.0; q[k] = qt[k]; }
  for (size_t k = tid; k < ns; k += 256) { a.so[b * ns + k] = 0; a.sl[b * ns + k] = 0; a.sc[b * ns + k] = 0.0; }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const long long s0 = a.sess_seg[b] - a.s_base, s1 = a.sess_seg[b + 1] - a.s_base;
  for (long long s = s0 + wave; s < s1; s += 4) {
    const int ev = a.s_evse[s], off = a.s_off[s], len = a.s_len[s];
    const long long r0 = a.rate_seg[s] - a.r_base;
    for (int p_ = lane; p_ < len; p_ += 64) {
      const double lo = a.min_rates[r0 + p_], hi = a.max_rates[r0 + p_];
      lb[(size_t)ev * a.Tm + off + p_] = lo;
      ub[(size_t)ev * a.Tm + off + p_] = hi < lo ? lo : hi;
    }
    if (lane == 0 && len > 0) {
      const size_t k = (size_t)b * ns + (size_t)a.s_slot[s] * a.N + ev;
      a.so[k] = off; a.sl[k] = len; a.sc[k] = a.s_cap[s];
    }
  }
}

int check_table(const acnqp_handle* h, const acnqp_table* t, const acnqp_options* o, const acnqp_results* r) {
  if (!h || !t || !o || !r) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: null argument");
  if (t->batch < 0) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: negative batch");
  if (t->batch == 0) return ACNQP_OK;
  if (!t->horizon || !t->q_index || !t->q_table || !t->pdiag || !t->s_eq || !t->sess_seg || !t->rate_seg ||
      (t->n_sessions > 0 && (!t->s_evse || !t->s_slot || !t->s_off || !t->s_len || !t->s_cap)))
    return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: null table array");
  if (t->n_sessions < 0 || t->n_horizons < 1) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: bad n_sessions / n_horizons");
  // the dense view for the shared checks (pointers only tested for null there)
  static const double dummy_d = 0;
  static const int32_t dummy_i = 0;
  acnqp_problems p;
  p.batch = t->batch; p.t_max = t->t_max; p.k_sessions = t->k_sessions; p.horizon = t->horizon;
  p.lb = p.ub = p.q = p.s_cap = &dummy_d; p.pdiag = t->pdiag; p.s_off = p.s_len = &dummy_i; p.s_eq = t->s_eq;
  p.peak = t->peak; p.lf = t->lf; p.dc = t->dc; p.dfloor = t->dfloor; p.warm_x = p.warm_y = nullptr;
  const int rc = check_problem_shapes(h, &p, o, r);
  if (rc != ACNQP_OK) return rc;
  const int B = t->batch, S = t->n_sessions, N = h->N, Tm = t->t_max, K = t->k_sessions;
  if (t->sess_seg[0] != 0 || t->sess_seg[B] != S) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: sess_seg must run from 0 to n_sessions");
  for (int b = 0; b < B; ++b) {
    if (t->sess_seg[b + 1] < t->sess_seg[b]) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: sess_seg is not non-decreasing");
    if (t->q_index[b] < 0 || t->q_index[b] >= t->n_horizons) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: q_index out of range");
    if (t->horizon[b] < 1 || t->horizon[b] > Tm) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: horizon out of range");
  }
  if (t->rate_seg[0] != 0) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: rate_seg must start at 0");
  for (int s_ = 0; s_ < S; ++s_) {
    const int len = t->s_len[s_] > 0 ? t->s_len[s_] : 0;
    if (t->s_evse[s_] < 0 || t->s_evse[s_] >= N || (len > 0 && (t->s_slot[s_] < 0 || t->s_slot[s_] >= K)))
      return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: session " + std::to_string(s_) + ": EVSE or slot out of range");
    if (len > 0 && (t->s_off[s_] < 0 || t->s_off[s_] + len > Tm))
      return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: session " + std::to_string(s_) + ": window outside [0, t_max)");
    if (t->rate_seg[s_ + 1] - t->rate_seg[s_] != len)
      return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: session " + std::to_string(s_) +
                                     ": min_rates / max_rates must have one entry per remaining period (aco.py:68, 73)");
  }
  if (t->rate_seg[S] > 0 && (!t->min_rates || !t->max_rates)) return fail(ACNQP_ERR_INVALID, "acnqp_solve_table: null rate arrays");
  return ACNQP_OK;
}

int run_table_pipeline(acnqp_handle* h, const acnqp_table* T, const acnqp_options* o, acnqp_results* R) {
  const size_t N = h->N, Tm = T->t_max, K = T->k_sessions, Mg = (size_t)h->Mg;
  const bool peak = h->has_peak, flat = h->has_flat, mx = h->has_max, want_y = R->y != nullptr;
  const long long B = T->batch;
  const size_t nv = N * Tm, nsl = K * N;
  long long cap0 = chunk_problems(4 * nv * 8 + nsl * 16 + Tm * 8 + 96, tiled_shape(h, (int)Tm, (int)K),
                                  acnqp::wave_shape(h->N, (int)Tm, (int)K, h->dev64.MR, h->has_max, (int)std::min<long long>(B, 1 << 30)) > 0);
  static const bool ramp = std::getenv("ACNQP_NO_RAMP") == nullptr;
  const bool is_wave = acnqp::wave_shape(h->N, (int)Tm, (int)K, h->dev64.MR, h->has_max, (int)std::min<long long>(B, 1 << 30)) > 0;
  std::vector<long long> plan;
  if (ramp && is_wave) plan_chunks(B, cap0, &plan);
  long long lo = 0;
  for (size_t c = 0; lo < B; ++c) {
    long long cap = cap0;
    if (!plan.empty()) cap = c < plan.size() ? plan[c] : cap0;
    else if (ramp && cap0 >= 1024) cap = c == 0 ? cap0 / 4 : (c == 1 ? cap0 / 2 : cap0);
    const long long cn = std::min(B - lo, cap);
    acnqp_handle::Slot& S = h->slot[c % acnqp_handle::kSlots];
    const ChunkLayout L((size_t)cn, N, Tm, K, Mg, peak, flat, mx, false, want_y);
    const long long s0 = T->sess_seg[lo], s1 = T->sess_seg[lo + cn], ns = s1 - s0;
    const long long r0 = T->rate_seg[s0], r1 = T->rate_seg[s1], nr = r1 - r0;
    // table staging: q_index[cn], sess_seg[cn + 1], {evse, slot, off, len}[ns], rate_seg[ns + 1], s_cap[ns], min / max [nr], q_table
    size_t to = 0;
    const size_t o_qi = to; to += al256((size_t)cn * 4);
    const size_t o_sg = to; to += al256((size_t)(cn + 1) * 4);
    const size_t o_ev = to; to += al256((size_t)ns * 4);
    const size_t o_sl = to; to += al256((size_t)ns * 4);
    const size_t o_of = to; to += al256((size_t)ns * 4);
    const size_t o_ln = to; to += al256((size_t)ns * 4);
    const size_t o_rs = to; to += al256((size_t)(ns + 1) * 4);
    const size_t o_cp = to; to += al256((size_t)ns * 8);
    const size_t o_mn = to; to += al256((size_t)nr * 8);
    const size_t o_mxr = to; to += al256((size_t)nr * 8);
    const size_t o_qt = to; to += al256((size_t)T->n_horizons * nv * 8);
    if (L.in_total > S.in.cap || L.out_total > S.out.cap || to > S.tin.cap) HIP_TRY(hipStreamSynchronize(S.st));   // staging still in use
    HIP_TRY(S.in.reserve(L.in_total));
    HIP_TRY(S.out.reserve(L.out_total));
    HIP_TRY(S.tin.reserve(to));
    char* di = static_cast<char*>(S.in.p);
    char* dq = static_cast<char*>(S.out.p);
    char* dt = static_cast<char*>(S.tin.p);
#define TH2D(dst, src, bytes) do { if ((bytes) > 0) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, S.st)); } while (0)
    TH2D(di + L.hz, T->horizon + lo, (size_t)cn * 4);
    TH2D(di + L.pd, T->pdiag + lo, (size_t)cn * 8);
    TH2D(di + L.eq, T->s_eq + lo, (size_t)cn);
    if (peak) TH2D(di + L.pk, T->peak + lo * Tm, (size_t)cn * Tm * 8);
    if (flat) TH2D(di + L.lf, T->lf + lo, (size_t)cn * 8);
    if (mx) { TH2D(di + L.dc, T->dc + lo, (size_t)cn * 8); TH2D(di + L.df, T->dfloor + lo, (size_t)cn * 8); }
    TH2D(dt + o_qi, T->q_index + lo, (size_t)cn * 4);
    TH2D(dt + o_sg, T->sess_seg + lo, (size_t)(cn + 1) * 4);
    TH2D(dt + o_ev, T->s_evse + s0, (size_t)ns * 4);
    TH2D(dt + o_sl, T->s_slot + s0, (size_t)ns * 4);
    TH2D(dt + o_of, T->s_off + s0, (size_t)ns * 4);
    TH2D(dt + o_ln, T->s_len + s0, (size_t)ns * 4);
    TH2D(dt + o_rs, T->rate_seg + s0, (size_t)(ns + 1) * 4);
    TH2D(dt + o_cp, T->s_cap + s0, (size_t)ns * 8);
    TH2D(dt + o_mn, T->min_rates + r0, (size_t)nr * 8);
    TH2D(dt + o_mxr, T->max_rates + r0, (size_t)nr * 8);
    TH2D(dt + o_qt, T->q_table, (size_t)T->n_horizons * nv * 8);
#undef TH2D
    TableExpandArgs ea;
    ea.N = (int)N; ea.Tm = (int)Tm; ea.K = (int)K; ea.s_base = s0; ea.r_base = r0;
    ea.q_index = reinterpret_cast<const int32_t*>(dt + o_qi); ea.sess_seg = reinterpret_cast<const int32_t*>(dt + o_sg);
    ea.s_evse = reinterpret_cast<const int32_t*>(dt + o_ev); ea.s_slot = reinterpret_cast<const int32_t*>(dt + o_sl);
    ea.s_off = reinterpret_cast<const int32_t*>(dt + o_of); ea.s_len = reinterpret_cast<const int32_t*>(dt + o_ln);
    ea.rate_seg = reinterpret_cast<const int32_t*>(dt + o_rs); ea.q_table = reinterpret_cast<const double*>(dt + o_qt);
    ea.s_cap = reinterpret_cast<const double*>(dt + o_cp); ea.min_rates = reinterpret_cast<const double*>(dt + o_mn);
    ea.max_rates = reinterpret_cast<const double*>(dt + o_mxr);
    ea.lb = reinterpret_cast<double*>(di + L.lb); ea.ub = reinterpret_cast<double*>(di + L.ub); ea.q = reinterpret_cast<double*>(di + L.q);
    ea.so = reinterpret_cast<int32_t*>(di + L.so); ea.sl = reinterpret_cast<int32_t*>(di + L.sl); ea.sc = reinterpret_cast<double*>(di + L.sc);
    hipLaunchKernelGGL(table_expand_kernel, dim3((unsigned)cn), dim3(256), 0, S.st, ea);
    acnqp_problems dp;
    dp.batch = (int32_t)cn; dp.t_max = (int32_t)Tm; dp.k_sessions = (int32_t)K;
    dp.lb = ea.lb; dp.ub = ea.ub; dp.q = ea.q;
    dp.pdiag = reinterpret_cast<const double*>(di + L.pd);
    dp.horizon = reinterpret_cast<const int32_t*>(di + L.hz);
    dp.s_off = ea.so; dp.s_len = ea.sl; dp.s_cap = ea.sc;
    dp.s_eq = reinterpret_cast<const uint8_t*>(di + L.eq);
    dp.peak = peak ? reinterpret_cast<const double*>(di + L.pk) : nullptr;
    dp.lf = flat ? reinterpret_cast<const double*>(di + L.lf) : nullptr;
    dp.dc = mx ? reinterpret_cast<const double*>(di + L.dc) : nullptr;
    dp.dfloor = mx ? reinterpret_cast<const double*>(di + L.df) : nullptr;
    dp.warm_x = nullptr; dp.warm_y = nullptr;
    acnqp_results dr;
    dr.x = reinterpret_cast<double*>(dq + L.x);
    dr.status = reinterpret_cast<int32_t*>(dq + L.st);
    dr.iters = reinterpret_cast<int32_t*>(dq + L.it);
    dr.pri_res = reinterpret_cast<double*>(dq + L.pr);
    dr.dua_res = reinterpret_cast<double*>(dq + L.du);
    dr.obj = reinterpret_cast<double*>(dq + L.ob);
    dr.x_dev = nullptr;
    dr.y = want_y ? reinterpret_cast<double*>(dq + L.y) : nullptr;
    const int rc = acnqp_solve_batch_device(h, &dp, o, &dr, S.st);
    if (rc != ACNQP_OK) return rc;
#define TD2H(field, base, elem, per)                                                                                          \
  HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(R->field) + (size_t)lo * (per) * (elem), dq + (base), (size_t)cn * (per) * (elem), \
                         hipMemcpyDeviceToHost, S.st))
    TD2H(x, L.x, 8, nv);
    TD2H(status, L.st, 4, 1);
    TD2H(iters, L.it, 4, 1);
    TD2H(pri_res, L.pr, 8, 1);
    TD2H(dua_res, L.du, 8, 1);
    TD2H(obj, L.ob, 8, 1);
    if (want_y) TD2H(y, L.y, 8, Mg * Tm);
#undef TD2H
    if (R->x_dev) HIP_TRY(hipMemcpyAsync(R->x_dev + (size_t)lo * nv, dq + L.x, (size_t)cn * nv * 8, hipMemcpyDeviceToDevice, S.st));
    lo += cn;
  }
  return ACNQP_OK;
}

}  // namespace

int acnqp_solve_table(acnqp_handle* h, const acnqp_table* t, const acnqp_options* o, acnqp_results* r) {
  const auto tcheck0 = std::chrono::steady_clock::now();
  const int rc0 = check_table(h, t, o, r);
  if (rc0 != ACNQP_OK) return rc0;
  if (t->batch == 0) return ACNQP_OK;
  HIP_TRY(hipSetDevice(h->device));
  static const bool trace = std::getenv("ACNQP_TRACE") != nullptr;   // diagnostic: check / enqueue / whole call, on stderr
  const auto tc0 = std::chrono::steady_clock::now();
  const int rc = run_table_pipeline(h, t, o, r);
  const auto tc1 = std::chrono::steady_clock::now();
  hipError_t e = hipSuccess;   // drain every slot before returning, also on failure: nothing may touch the caller's buffers afterwards
  for (auto& sl : h->slot) { const hipError_t e1 = hipStreamSynchronize(sl.st); if (e == hipSuccess) e = e1; }
  if (trace) {
    const auto tc2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[acnqp] solve_table: check %.3f ms, enqueue %.3f ms, drained after %.3f ms\n",
                 std::chrono::duration<double, std::milli>(tc0 - tcheck0).count(), std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
                 std::chrono::duration<double, std::milli>(tc2 - tc0).count());
  }
  if (rc != ACNQP_OK) return rc;
  if (e != hipSuccess) return fail(ACNQP_ERR_HIP, std::string("acnqp_solve_table: ") + hipGetErrorString(e));
  for (int b = 0; b < t->batch; ++b)
    if (r->status[b] == ACNQP_STATUS_UNSET)
      return fail(ACNQP_ERR_HIP, "acnqp_solve_table: problem " + std::to_string(b) +
                                 " was never written by the kernel (status UNSET after synchronisation): the launch did not execute completely");
  return ACNQP_OK;
}

int acnqp_solve_batches(acnqp_handle* h, int32_t n_batches, const acnqp_problems* p, const acnqp_options* o, acnqp_results* r) {
  if (!h || !p || !o || !r || n_batches < 0) return fail(ACNQP_ERR_INVALID, "acnqp_solve_batches: null argument or negative count");
  for (int g = 0; g < n_batches; ++g) {
    const int rc = check_problem_shapes(h, p + g, o, r + g);
    if (rc != ACNQP_OK) return rc;
  }
  HIP_TRY(hipSetDevice(h->device));
  static const bool trace = std::getenv("ACNQP_TRACE") != nullptr;   // diagnostic: host enqueue time vs the whole call, on stderr
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<Scatter> scatter;
  const int rc = run_pipeline(h, n_batches, p, o, r, &scatter);
  const auto t1 = std::chrono::steady_clock::now();
  // drain every slot before returning, also on failure: nothing may touch the caller's buffers afterwards
  hipError_t e = hipSuccess;
  for (auto& sl : h->slot) { const hipError_t e1 = hipStreamSynchronize(sl.st); if (e == hipSuccess) e = e1; }
  if (trace) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[acnqp] solve_batches: enqueue %.3f ms, drained after %.3f ms\n",
                 std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
  }
  if (rc != ACNQP_OK) return rc;
  if (e != hipSuccess) return fail(ACNQP_ERR_HIP, std::string("acnqp_solve_batches: ") + hipGetErrorString(e));
  for (const Scatter& sc : scatter) {   // the small result arrays: pinned mirror -> the caller's arrays
    const acnqp_results& rr = r[sc.g];
    std::memcpy(rr.status + sc.lo, sc.host + sc.st + sc.pos * 4, sc.n * 4);
    std::memcpy(rr.iters + sc.lo, sc.host + sc.it + sc.pos * 4, sc.n * 4);
    std::memcpy(rr.pri_res + sc.lo, sc.host + sc.pr + sc.pos * 8, sc.n * 8);
    std::memcpy(rr.dua_res + sc.lo, sc.host + sc.du + sc.pos * 8, sc.n * 8);
    std::memcpy(rr.obj + sc.lo, sc.host + sc.ob + sc.pos * 8, sc.n * 8);
  }
  for (int g = 0; g < n_batches; ++g)
    for (int b = 0; b < p[g].batch; ++b)
      if (r[g].status[b] == ACNQP_STATUS_UNSET)
        return fail(ACNQP_ERR_HIP, "acnqp_solve_batches: batch " + std::to_string(g) + " problem " + std::to_string(b) +
                                   " was never written by the kernel (status UNSET after synchronisation): the launch did not execute completely");
  return ACNQP_OK;
}

int acnqp_solve_batch(acnqp_handle* h, const acnqp_problems* p, const acnqp_options* o, acnqp_results* r) {
  return acnqp_solve_batches(h, 1, p, o, r);
}

void* acnqp_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    g_last_error = "acnqp_host_alloc: hipHostMalloc failed";
    return nullptr;
  }
  return p;
}

void acnqp_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
