// MFMA-tiled batched MPC QP solver for gfx950: one workgroup per problem, one wavefront per
// 16-EVSE tile, every cross-EVSE product as a chain of v_mfma_f64_16x16x4_f64.
//
// Device algorithm (restated on the CPU in oracle/admm_ref.py and oracle/admm_port.c):
//   x~   = (a I + rho G'G)^-1 (sigma x - q + rho z1 - y1 + G'(rho z2 - y2))   per period
//   z1   = Proj_B (alpha x~ + (1-alpha) z1 + y1/rho),   B = box /\ session energy rows
//   z2   = Proj_C (alpha G x~ + (1-alpha) z2 + y2/rho), C = site rows (box / disc / peak)
//   y    = rho (pre-projection point - projected point)
// with (a I + rho G'G)^-1 = (I - Ghat' D Ghat)/a,  G G' = Q Lam Q',  Ghat = Q'G,  D = rho/(a + rho Lam).
//
// Layout.  Wave w, lane l = 16 g + t (g = l >> 4, t = l & 15) holds, for each column tile c and
// register r, the entry (EVSE 16 w + rowof(g, r), period 16 c + t) of every N x T iterate: exactly
// the C/D operand map of the 16x16x4 MFMA (rowof = g + 4 r for f64, 4 g + r for f32), so that
//   * the wave's own r0 registers are the B operand of  P_w = Ghat[:, tile w] r0_w          (phase 1)
//   * x~ tile = r0 + Ghat[:, tile w]' e^ comes out of the MFMA in the layout it is consumed in
//   * an accumulator tile feeds the next chain as B operand (register s <-> k-step s), so
//     w^ = Q'w, e^, h^ and G x~ = Q h^ never leave registers.
// The site-row space (<= 48 rows) is tiny, so every wave keeps its own copy of (z2, y2, G x) and
// updates it redundantly: the only exchange per iteration is the sum of the NW partial tiles P_w
// through a double-buffered LDS slab, i.e. ONE workgroup barrier per iteration.  A session's
// energy sum runs over the 16 lanes of a row: four DPP steps (quad_perm, quad_perm,
// row_half_mirror, row_mirror), no LDS.  HBM is touched once per problem.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acnqp {

#ifdef ACNQP_STAMPS
// diagnostic build only: cycles per phase, [block][wave][8]; never read by the kernel
__device__ unsigned long long g_stamps[1024 * 16 * 12];
#define STAMP(slot)                                                                    \
  do {                                                                                 \
    unsigned long long _t;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    st_acc[slot] += _t - st_prev;                                                      \
    st_prev = _t;                                                                      \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

#ifndef ACNQP_GUARD_MAX
#define ACNQP_GUARD_MAX 120   // safety bound on water-filling passes per session and iteration
#endif

constexpr int kMaxK = 4;       // session slots per EVSE
constexpr int kStatusPolish = 6;   // internal status: "left to the polish kernel" (never returned to a caller)
constexpr int kNumRed = 8;

// row types of the (internally ordered) site rows
constexpr int kRowFree = 0;    // padding row: no constraint
constexpr int kRowBox = 1;     // z <= limit
constexpr int kRowSocRe = 2;   // pairs with the next register (kRowSocIm): |(re, im)| <= limit
constexpr int kRowSocIm = 3;
constexpr int kRowPeak = 4;    // z <= peak[b][t]
constexpr int kRowMax = 6;     // prox of dc * max(max_t z_t, floor) over the whole horizon (demand charge)
constexpr int kRowQuad = 5;    // prox of 1/2 lf z^2 (load flattening): z = zh rho / (rho + lf)

struct TiledArgs {
  int B, N, Tm, K, NP, MR;     // NP = 16 * waves (padded EVSEs), MR = 16 * MT (padded site rows)
  const void *G, *Ghat, *Q, *lam, *rowlim;   // [MR][NP], [MR][NP], [MR][MR], [MR], [MR]  (real)
  const int32_t* rowtype;                     // [MR]
  const void *fragG, *fragQ;                  // Ghat, Q as MFMA A-operand fragments: [NW][MT][2][4][64], [MT][MT][2][4][64]
  const void *fragG2, *fragQ2;                // the same blocks with the k-slices as two pairs per lane: [..][2][64][2] (acn_qp_long.hpp)
  const int32_t* horizon;
  const int32_t* order;                       // [B] or null: queue position -> problem (longest expected first, acn_qp_api.hip)
  int32_t* queue;                             // launch counter (zeroed on the stream before the launch): the grid is the chip's
                                              // resident workgroup slots and every workgroup fetches its next queue position
                                              // with one atomicAdd until the B problems are gone (null: workgroup w solves
                                              // position w -- ACNQP_NO_QUEUE=1, the static schedule)
  const double *lb, *ub, *q, *pdiag;
  const int32_t *s_off, *s_len;
  const double* s_cap;
  const uint8_t* s_eq;
  const double* peak;
  const double* lf;
  const double *dc, *dfloor;
  const double *warm_x, *warm_y;   // optional warm start (both or neither): a schedule [B][N][Tm] and site-row multipliers
                                   // [B][Mg][Tm] in the row order and units of acnqp_site.G
  const int32_t* rowabi;           // [MR] internal row -> row of acnqp_site.G (-1: padding)
  const void* rowscale;            // [MR] (real) equilibration factor of the internal row: y_abi = scale * y_internal
  int Mg;                          // rows of acnqp_site.G
  double* x;
  double* y_out;                   // optional: the site-row multipliers at exit, [B][Mg][Tm]
  int32_t *status, *iters;
  double *pri, *dua, *obj;
  double eps_abs, eps_rel, rho0, sigma, alpha, adapt_tol, reg_rel;
  double peak_scale, flat_scale, max_scale;   // host-side row equilibration of the prox rows
  int max_iter, check_every, adapt_every;
  int accel_mem;   // Anderson-acceleration columns actually used (<= the kernel's AM, fits its LDS); 0 = off
  // acnqp_options.stall_iters / inaccurate_floor / retry_* (include/acn_qp.h)
  int stall_iters, retry_passes, retry_max_iter;
  double inacc_floor, retry_rho;
  int pbuf_single; // 1: one partial-tile slab instead of two (one more barrier per iteration, LDS for one more ring column)
  // -- the polish (acn_qp_polish.hpp): a pass-0 problem that has not converged after polish_iters iterations (0 = off) leaves
  //    the solver kernel with the internal status kStatusPolish, its iterate in x / y_out, and its index appended to pol_list;
  //    `resume` = 1 is the launch that follows the polish kernel: it runs over pol_list (order = pol_list, count_dev = its
  //    length, on the device), skips what the polish solved and solves the rest from scratch as if there were no polish
  int polish_iters, resume;
  int polish_stall;           // > 0: from polish_iters / 2 on, a problem whose residual score has not improved by 10 % for this many
                              // iterations is handed over early (acn_qp_wave.hpp; acn_qp_api.hip sets polish_iters / 4)
  int y_for_polish_only;      // 1: y_out is the polish's internal buffer -- only a problem that is handed over writes it (the
                              // multipliers of every problem were 5.4 KB of HBM writes per problem for 1.5 KB of payload)
  int pol_rows;               // rows of the polish kernel's Schur system for this shape: a problem whose iterate has more tight site
                              // rows than that is not handed over (it would come straight back) and the ADMM goes on
  int32_t *pol_list, *pol_count;
  const int32_t* count_dev;   // number of queue positions, on the device (null: B)
  int ws_by_slot;  // 1: a streaming kernel's workspace belongs to the workgroup slot (work-queue launches), 0: to the problem
  int grid_oversub; // host side only: workgroups of a work-queue launch per resident slot (1: fully persistent; the pipelined
                    // host entries use 4, so that slots come free while a launch runs and the next stream's small launches --
                    // the polish, the resume -- do not wait for a whole persistent launch to drain)
  int grid_cap;    // host side only: most workgroups a launch may have (the kernels that stream their state own one
                   // workspace per workgroup slot)
};

typedef const __attribute__((address_space(4))) TiledArgs* KernargPtr;   // the kernel's own argument block

// ---- work queue (all four kernel families) ----------------------------------------------------------------------------
// A launch of B problems on S resident workgroup slots used to be B workgroups handed out in index order: a slot's
// share of the work was whatever its problems happened to need, and the launch ended with its slowest slot (configs[4]
// leg, 2,048 problems on 512 slots: the last slot at ~2,200 iterations, the mean slot at 1,525).  Now the grid is the S
// slots and a workgroup that finishes a problem takes the next queue position: one atomicAdd by thread 0, broadcast
// through LDS.  Returns the position (block-uniform, a scalar), or -1 when the launch has no problem left for this
// workgroup.  `round` counts this workgroup's fetches (the static fallback serves exactly one).  Both barriers are
// needed: the first publishes the slot, the second keeps a fast wave's next fetch from overwriting it before a slow wave
// has read it -- and orders the previous problem's last LDS reads before the next problem's first LDS writes.
__device__ inline int queue_length(const TiledArgs& a) {
  if (a.count_dev == nullptr) return a.B;
  const int n = *a.count_dev;
  return n < a.B ? n : a.B;
}
__device__ inline int queue_next(int32_t* queue, int B, int round, int* slot_lds) {
  if (queue == nullptr) return round == 0 && (int)blockIdx.x < B ? (int)blockIdx.x : -1;
  if (threadIdx.x == 0) *slot_lds = atomicAdd(queue, 1);
  __syncthreads();
  const int pos = __builtin_amdgcn_readfirstlane(*slot_lds);
  __syncthreads();
  return pos < B ? pos : -1;
}

template <typename real> struct Mfma;
template <> struct Mfma<double> {
  typedef double vec4 __attribute__((ext_vector_type(4)));
  __host__ __device__ static constexpr int rowof(int g, int r) { return g + 4 * r; }
  __device__ static inline vec4 mma(double a, double b, vec4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static constexpr double proj_tol = 1e-13;
  static constexpr double big = 1e300;
};

// ---- cross-lane helpers (DPP: plain VALU moves, no LDS crossbar) -------------------------------
template <int CTRL> __device__ inline float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ inline double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;

// reductions over the 16 lanes of a DPP row (= the 16 periods of one EVSE); every lane gets the result
template <typename T> __device__ inline T row_sum(T v) {
  v += dpp_mov<kQuadXor1>(v);
  v += dpp_mov<kQuadXor2>(v);
  v += dpp_mov<kRowHalfMirror>(v);
  v += dpp_mov<kRowMirror>(v);
  return v;
}
template <typename T> __device__ inline T row_min(T v) {
  v = fmin(v, dpp_mov<kQuadXor1>(v));
  v = fmin(v, dpp_mov<kQuadXor2>(v));
  v = fmin(v, dpp_mov<kRowHalfMirror>(v));
  v = fmin(v, dpp_mov<kRowMirror>(v));
  return v;
}
template <typename T> __device__ inline T row_max(T v) {
  v = fmax(v, dpp_mov<kQuadXor1>(v));
  v = fmax(v, dpp_mov<kQuadXor2>(v));
  v = fmax(v, dpp_mov<kRowHalfMirror>(v));
  v = fmax(v, dpp_mov<kRowMirror>(v));
  return v;
}

// A block-uniform double moved to the scalar unit (two v_readfirstlane): it then occupies a scalar register pair instead of
// two vector registers per lane for as long as it lives (the kernels' penalty, its reciprocals, norms, scores ...).
__device__ inline double uniform_scalar(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// A double constant materialised in a scalar register pair where it is used (two s_mov_b32), opaque to the optimiser:
// without this the compiler hoists every such constant of the kernel (1e300, 1.2, 0.9, 1e-300, 1e-6 ...) into a vector
// register pair at the top and keeps -- or spills -- it across the solver loop.
__device__ inline double scalar_const(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// The value lane `src` holds (src uniform), as a scalar: v_readlane instead of the LDS crossbar a __shfl goes through
__device__ inline double lane_value(double v, int src) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, src), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline float uniform_scalar(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, v))); }

// 1 / sqrt(x): hardware estimate + two Newton steps (full precision for f64, cheaper than sqrt + div)
__device__ inline double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ inline float rsqrt_nr(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  y = y * (1.5f - 0.5f * x * y * y);
  return y;
}

// 1 / x for x > 0: hardware estimate + Newton steps (avoids the long IEEE division sequence)
__device__ inline double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y + y * (1.0 - x * y);
  y = y + y * (1.0 - x * y);
  return y;
}
__device__ inline float rcp_nr(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  y = y + y * (1.0f - x * y);
  return y;
}

// 1 / n for a small positive integer n (interior-period count of a session window)
__device__ inline double rcp_small(float nf) {
  const double x = (double)nf;
  double y = (double)__builtin_amdgcn_rcpf(nf);   // ~1e-7 relative
  y = y + y * (1.0 - x * y);                      // ~1e-14
  y = y + y * (1.0 - x * y);                      // full double precision
  return y;
}

// LDS carve-up in units of `real`; shared by host (size) and device (offsets)
constexpr int kXS = 18;   // row stride (reals) of the per-wave 16 x 16 transpose scratch: 16-B aligned quads
struct TiledLds {
  int pbuf, xpose, red, rowc, fragq, aared, aah, hist, snap, total;   // offsets in reals; hist..total hold floats
  int hist1, hist2;                                // floats per history column: tile part, site-row part
  int pstride;                                     // reals between the partial-tile regions of two waves in a slab
  __host__ __device__ TiledLds(int NW, int MT, int CT, int NP, int K, int AM, int accel_mem, int real_bytes, int pbuf_single = 0) {
    int o = 0;
    // Partial tiles: two slabs (one with pbuf_single), each NW per-wave regions.  With two slabs the wave's transpose
    // scratch (C layout <-> session layout, CT * 16 * kXS reals) ALIASES the wave's own region of the slab that is idle
    // in the current iteration -- everybody finished reading it before this iteration's barrier, and its next writer is
    // this very wave at the top of the next iteration -- so a region is as long as the longer of the two.
    const int ptile = MT * CT * 4 * 64, xp = CT * 16 * kXS;
    pstride = pbuf_single ? ptile : (ptile > xp ? ptile : xp);
    pbuf = o;  o += (pbuf_single ? 1 : 2) * NW * pstride;
    xpose = o; o += pbuf_single ? NW * xp : 0;   // a separate scratch only when there is no idle slab
    red = o;   o += 16 * kNumRed + 8;
    rowc = o;  o += 3 * 16 * MT + 8 * MT;        // per site row: eigenvalue, limit, rho / (a + rho lam); then the row types (ints)
    fragq = o; o += MT == 1 ? 2 * 4 * 64 : 0;    // one row tile: the Q fragments of the two site-row products (4 KB), read by every wave
    aared = o; o += accel_mem > 0 ? NW * (AM + 2) : 0;          // per-wave partial dot products
    aah = o;   o += accel_mem > 0 ? NW * (AM * AM + AM) : 0;    // per-wave copy of the Gram matrix and rhs
    o = (o + 1) & ~1;
    hist = o;
    hist1 = NW * 64 * CT * 4;
    hist2 = 64 * MT * CT * 4;
    const int hfloats = 2 * accel_mem * (hist1 + hist2);        // dF ring then dG ring
    o += (hfloats * 4 + real_bytes - 1) / real_bytes;
    o = (o + 1) & ~1;
    snap = o;                                                   // duals at the previous residual check (certificate),
    o += ((hist1 + hist2) * 4 + real_bytes - 1) / real_bytes;   // floats: [tid][CT*4], then ONE copy of the site part [lane][MT*CT*4]
    total = (o + 1) & ~1;
  }
  // bytes of LDS one Anderson column costs, and everything else (to size accel_mem on the host)
  __host__ static int column_bytes(int NW, int MT, int CT) { return 2 * (NW * 64 * CT * 4 + 64 * MT * CT * 4) * 4; }
};

// Anderson acceleration of the ADMM fixed-point map (restated in oracle/admm_port.c, see there)
constexpr double kStartGain = 1e5;
// Tikhonov floor (options.reg_rel) -- applied ONLY to problems whose own objective cannot select a unique point:
// no prox row in use (load_flattening / demand_charge weights zero) and a quadratic the solver cannot resolve,
// pdiag * max(ub) <= kRegResolve * |q|_inf (pure LPs; the reference's equal_share * 1e-12).  A strictly convex
// problem is solved exactly as stated (the reference passes no solver options, aco.py:315-321).
constexpr double kRegResolve = 1e-6;
template <typename real>
__host__ __device__ inline real effective_pdiag(real pd_user, real reg_rel, real qnorm, real ubmax, int horizon, bool has_prox) {
  if (has_prox || !(ubmax > (real)0) || pd_user * ubmax > (real)kRegResolve * qnorm) return pd_user;
  const real fl = reg_rel * qnorm / (ubmax * (real)(horizon > 1 ? horizon : 1));
  return fl > pd_user ? fl : pd_user;
}
// SOLVED_INACCURATE (the reference accepts cvxpy's OPTIMAL_INACCURATE, aco.py:319): when the iteration limit or the
// stall rule ends a problem with both residuals within kInaccurate x the requested tolerance, or within the tolerance
// cvxpy hands OSQP by default (eps_abs = eps_rel = 1e-5), whichever is looser.
constexpr double kInaccurate = 100.0;
template <typename real>
__host__ __device__ inline bool inaccurate_ok(real pri, real dua, real npri, real ndua, double eps_abs, double eps_rel, double floor_) {
  const real ea = (real)(kInaccurate * eps_abs > floor_ ? kInaccurate * eps_abs : floor_);
  const real er = (real)(kInaccurate * eps_rel > floor_ ? kInaccurate * eps_rel : floor_);
  return pri <= ea + er * npri && dua <= ea + er * ndua;
}
// Retry passes (acnqp_options.retry_passes): a problem that ends a pass MAX_ITER / SOLVED_INACCURATE after at least
// retry_min_iters(stall_iters) iterations is solved again, inside the same kernel, from a cold start with a FIXED
// penalty retry_rho * 4^(pass - 1) (no adaptation) and at most retry_max_iter iterations.  Why: traced on the C twin
// (DESIGN.md section 2), the congested instances that sit on a plateau of the primal residual do so because the penalty
// adaptation swings rho by 10x several times in the first few hundred iterations and the iterate ends in a region it
// leaves only sub-linearly; from a cold start with a fixed rho in [0.3, 4] every one of them converges in 900 ... 4,400
// iterations -- while a fixed rho for everybody would double the iterations of the average problem.  The answer of
// the best pass is the one returned (SOLVED beats SOLVED_INACCURATE beats MAX_ITER; the first of equals), iters is
// the total over the passes.
__host__ __device__ inline int retry_min_iters(int stall_iters) { return stall_iters > 0 ? stall_iters : 3000; }
__host__ __device__ inline int status_rank(int st) { return st == 1 ? 3 : (st == 5 ? 2 : (st == 2 ? 1 : 0)); }
__host__ __device__ inline bool retry_wanted(int pass, int retry_passes, int status, int it, int stall_iters, int adapt_every0) {
  return pass < retry_passes && (status == 2 || status == 5) && it >= retry_min_iters(stall_iters) && adapt_every0 > 0;
}
// Stall rule: a problem whose residual score max(pri / eps_pri, dua / eps_dua) has not improved by 10 % for
// kStallIters iterations and sits within kStallNear of its best score (i.e. on the plateau, not in the transient after
// a rho change) is finished: SOLVED_INACCURATE if it qualifies by the rule above, MAX_ITER otherwise (what it would be
// max_iter - it iterations later; the binding's second pass re-solves both kinds).  Converging problems never
// wait that long between improvements (longest wait seen on solved instances of every shape in tools/ and tests/:
// 1,240 iterations, a caltech54 x 12 LINEAR LP); the ones that do are the tangentially degenerate congested instances of DESIGN.md section 6, which
// otherwise burn max_iter iterations on a plateau and end with the same status.
constexpr double kStallGain = 0.9, kStallNear = 1.25;   // the window is acnqp_options.stall_iters (default 3000, 0 = off)
constexpr double kAdaptWiden = 8.0;   // rho adaptation band: adapt_tol (1 + adaptations / kAdaptWiden): no limit cycles
constexpr int kAaPeriod = 5;
constexpr double kAaReg = 1e-4, kAaSafe = 1.2, kAaDrift = 1e-3;

// Reductions over the lanes l, l^16, l^32, l^48 (the four quarter-lanes of one EVSE in session
// layout) with v_permlane16_swap / v_permlane32_swap: swapping a register with itself leaves the
// two partner values in the two results for EVERY lane, so r[0] (op) r[1] is the pairwise result --
// plain VALU, no LDS crossbar, and bitwise identical in both partners.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct Pair32 { unsigned a, b; };
struct PairF { float a, b; };
struct PairD { double a, b; };
template <int WHICH> __device__ inline Pair32 swap_u32(unsigned v) {
  const u32x2 r = WHICH == 16 ? __builtin_amdgcn_permlane16_swap(v, v, false, false)
                              : __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return {r[0], r[1]};
}
template <int WHICH> __device__ inline PairF swap_pair(float v) {
  const Pair32 p = swap_u32<WHICH>(__builtin_bit_cast(unsigned, v));
  return {__builtin_bit_cast(float, p.a), __builtin_bit_cast(float, p.b)};
}
template <int WHICH> __device__ inline PairD swap_pair(double v) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const Pair32 lo = swap_u32<WHICH>((unsigned)(bits & 0xffffffffull));
  const Pair32 hi = swap_u32<WHICH>((unsigned)(bits >> 32));
  return {__builtin_bit_cast(double, ((unsigned long long)hi.a << 32) | lo.a),
          __builtin_bit_cast(double, ((unsigned long long)hi.b << 32) | lo.b)};
}
template <typename T> __device__ inline T quarter_sum(T v) {
  auto p = swap_pair<16>(v); v = p.a + p.b;
  auto q = swap_pair<32>(v); return q.a + q.b;
}
template <typename T> __device__ inline T quarter_min(T v) {
  auto p = swap_pair<16>(v); v = fmin(p.a, p.b);
  auto q = swap_pair<32>(v); return fmin(q.a, q.b);
}
template <typename T> __device__ inline T quarter_max(T v) {
  auto p = swap_pair<16>(v); v = fmax(p.a, p.b);
  auto q = swap_pair<32>(v); return fmax(q.a, q.b);
}

template <typename T> __device__ inline T wave_max(T v) { return quarter_max<T>(row_max<T>(v)); }
template <typename T> __device__ inline T wave_sum(T v) { return quarter_sum<T>(row_sum<T>(v)); }

// Block-wide max of NV per-thread values (2 barriers); every thread gets the result.
template <typename real, int NV>
__device__ inline void block_max(real (&v)[NV], real* Red, int lane, int wave, int nw) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const real m = wave_max<real>(v[k]);
    if (lane == 0) Red[wave * kNumRed + k] = m;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    real m = Red[k];
    for (int wv = 1; wv < nw; ++wv) m = fmax(m, Red[wv * kNumRed + k]);
    v[k] = m;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
template <typename real, int NW, int CT, int MT, int KS, int OCC, int AM>
__global__ __launch_bounds__(NW * 64, OCC) void admm_tiled_kernel(const TiledArgs A_kernarg) {
  using M = Mfma<real>;
#define BIGC ((real)scalar_const((double)M::big))
  using vec4 = typename M::vec4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* sm = reinterpret_cast<real*>(smem_raw);

  // ---- passes: pass 0 is the solve as the options state it; a problem it leaves MAX_ITER / SOLVED_INACCURATE on a
  // plateau is solved again from a cold start with a fixed penalty (retry_wanted, above).  The WHOLE body, loads
  // included, is the pass: with the thread and block ids opaque at the top nothing of a pass is invariant across
  // passes, so the compiler keeps no pass-invariant value (addresses, predicates) alive across the solver loop.
  __shared__ int q_slot;
  for (int q_round = 0;; ++q_round) {   // work queue: this workgroup's next problem (queue_next)
  const int q_pos = queue_next(A_kernarg.queue, queue_length(A_kernarg), q_round, &q_slot);
  if (q_pos < 0) break;
  int it_total = 0, best_status = 0;
  for (int pass = 0;; ++pass) {
  int b_ = q_pos;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(b_));
  asm volatile("" : "+v"(tid));
  const int wg_ = __builtin_amdgcn_readfirstlane(b_);
  // ... and the argument block is read through a per-pass opaque pointer to the kernarg segment (the by-value struct
  // sits at its offset 0): scalar loads from constant memory, hoisted freely inside a pass, never across passes
  // (otherwise every argument the loads and the start use stays in a scalar register through the solver loop)
  KernargPtr Ap = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(Ap));
  const auto& A = *Ap;
  (void)A_kernarg;
  const int b = __builtin_amdgcn_readfirstlane(A.order ? A.order[wg_] : wg_);   // the problem this workgroup solves (a uniform value: the load alone would make it a vector register)
  if (A.resume) {   // the launch behind the polish kernel: what it solved is done; the rest starts over (block-uniform)
    if (A.status[b] != kStatusPolish && pass == 0) break;
    if (pass == 0) it_total = A.iters[b];
  }
  const int max_iter_p = pass == 0 ? A.max_iter : min(A.max_iter, A.retry_max_iter);
  const int adapt_p = pass == 0 ? A.adapt_every : 0;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave index as a SCALAR: `if (wave == 0)` is a scalar branch,
                                                                                // per-wave LDS regions are scalar offsets (it was a vector value: divergent regions)
  const int g = lane >> 4, t = lane & 15;
  const int N = A.N, Tm = A.Tm, NP = A.NP, MR = A.MR;
  const int aa_m = AM > 0 ? min(A.accel_mem, AM) : 0;
  const TiledLds L(NW, MT, CT, NP, A.K, AM, aa_m, (int)sizeof(real), A.pbuf_single);
  real* Pbuf = sm + L.pbuf;
  const int PS = L.pstride;                                 // reals between two waves' regions of a slab
  // this wave's transpose scratch: its region of the idle slab (re-pointed at the top of every iteration; the start
  // stores its partial tiles in slab 0, so slab 1 is the idle one there), or a scratch of its own with one slab
  real* Xw = A.pbuf_single ? sm + L.xpose + (size_t)wave * CT * 16 * kXS : Pbuf + (size_t)NW * PS + (size_t)wave * PS;
  float* Y1P = reinterpret_cast<float*>(sm + L.snap) + (size_t)tid * (CT * 4);                 // certificate snapshot, own slot
  float* Y2P = reinterpret_cast<float*>(sm + L.snap) + (size_t)L.hist1 + (size_t)lane * (MT * CT * 4);   // site part: ONE copy, written by wave 0
  real* Red = sm + L.red;
  const real* Gm = static_cast<const real*>(A.G);
  const real* Gh = static_cast<const real*>(A.Ghat);
  const real* Qm = static_cast<const real*>(A.Q);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);

  // ---- MFMA A-operand fragments of this wave: re-read every iteration from the (L1/L2-resident, shared)
  // fragment-ordered copies of Ghat and Q (coalesced: one 64-lane row per fragment register) instead of
  // living in 32 registers per lane across the whole loop
  //   aP[m][s]      = Ghat[16m + t'][16w + rowof(g,s)]      (t' = lane & 15 is the A row)
  //   aX[m][s]      = Ghat[16m + rowof(g,s)][16w + t']
  //   aQt[mo][mi][s] = Q[16mi + rowof(g,s)][16mo + t']      (w^ = Q' w)
  //   aQ[mo][mi][s]  = Q[16mo + t'][16mi + rowof(g,s)]      (G x~ = Q h^)
  // site-row constants in C layout: row j = 16 m + rowof(g, r)
  // per-row constants (eigenvalue, limit, D = rho / (a + rho lam)) live in a small LDS table, not in 24 registers
  real* RowLam = sm + L.rowc;
  real* FQs = sm + L.fragq;   // MT == 1: Q fragments in LDS (an LDS read instead of a global load in front of 8 of the 16 MFMA)
  real* RowLim = RowLam + 16 * MT;
  real* RowDj = RowLim + 16 * MT;
  // row types: an LDS table in the lanes' own order -- entry (m, g, r) = type of row 16 m + rowof(g, r), the four of a
  // lane adjacent (one 16-byte read where they are needed) -- instead of four registers per row tile across the loop
  int* RowTy = reinterpret_cast<int*>(RowDj + 16 * MT);
  for (int j = tid; j < 16 * MT; j += NW * 64) {
    RowLam[j] = Lm[j]; RowLim[j] = RL[j];
    const int m_ = j >> 4, g_ = (j >> 2) & 3, r_ = j & 3;
    RowTy[j] = A.rowtype[16 * m_ + M::rowof(g_, r_)];
  }
  if constexpr (MT == 1) {   // (a barrier follows before the first use)
    const real* FQg = static_cast<const real*>(A.fragQ);
    for (int j = tid; j < 2 * 4 * 64; j += NW * 64) FQs[j] = FQg[j];
  }
  auto row_types = [&](int m, int (&ty)[4]) __attribute__((always_inline)) {
    const int4 v = *reinterpret_cast<const int4*>(RowTy + (m * 4 + g) * 4);
    ty[0] = v.x; ty[1] = v.y; ty[2] = v.z; ty[3] = v.w;
  };

  // ---- problem data -> registers (C layout) --------------------------------------------
  real x[CT][4], z1[CT][4], y1[CT][4], qv[CT][4], lbv[CT][4], ubv[CT][4];
  // the peak limit of the lane's period (scaled like its row), fetched where a peak row is projected
  auto peak_at = [&](int c) __attribute__((always_inline)) -> real {
    const int tt = 16 * c + t;
    double pv = 1e300;
    if (A.peak && tt < Tm) pv = A.peak[(size_t)b * Tm + tt];
    return pv < (double)BIGC ? (real)(pv * A.peak_scale) : BIGC;
  };
  bool evact[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) evact[r] = 16 * wave + M::rowof(g, r) < N;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int tt = 16 * c + t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ev = 16 * wave + M::rowof(g, r);
      const bool ok = evact[r] && tt < Tm;
      const size_t idx = ((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0);
      lbv[c][r] = ok ? (real)A.lb[idx] : (real)0;
      ubv[c][r] = ok ? (real)A.ub[idx] : (real)0;
      qv[c][r] = ok ? (real)A.q[idx] : (real)0;
      if (ubv[c][r] < lbv[c][r]) ubv[c][r] = lbv[c][r];
      x[c][r] = 0; z1[c][r] = 0; y1[c][r] = 0;
    }
  }
  const bool eq = A.s_eq[b] != 0;
  // ---- session layout: lane = 16 h + e holds periods 16 c + 4 h + tt (tt = 0..3) of EVSE 16 w + e.
  // The energy rows are solved here: a session's sum is 4 local adds + two lane exchanges, and one
  // pass over g(m) serves the wave's 16 sessions at once.
  const int se = lane & 15, sh = lane >> 4;
  const int sev = 16 * wave + se;
  const bool sact = sev < N;
  real slb[CT][4], sub[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int tp = 16 * c + 4 * sh + tt;
      const bool ok = sact && tp < Tm;
      const size_t idx = ((size_t)b * N + (ok ? sev : 0)) * Tm + (ok ? tp : 0);
      slb[c][tt] = ok ? (real)A.lb[idx] : (real)0;
      sub[c][tt] = ok ? (real)A.ub[idx] : (real)0;
      if (sub[c][tt] < slb[c][tt]) sub[c][tt] = slb[c][tt];
    }
  unsigned swm[KS];            // bit (4 c + tt): that period lies in slot k's window
  int smode[KS];               // 0: root-find each iteration, 2: pinned at ub, 3: pinned at lb, 4: no session in this slot
  real scap[KS], mu[KS];
  bool empty_set = false;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    swm[k] = 0; smode[k] = 4; scap[k] = 0; mu[k] = 0;
    if (k < A.K) {   // block-uniform
      const size_t sidx = ((size_t)b * A.K + k) * N + (sact ? sev : 0);
      const int off = sact ? A.s_off[sidx] : 0;
      const int len = sact ? A.s_len[sidx] : 0;
      real sl = 0, su = 0;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int tp = 16 * c + 4 * sh + tt;
          if (tp >= off && tp < off + len && tp < Tm) { swm[k] |= 1u << (4 * c + tt); sl += slb[c][tt]; su += sub[c][tt]; }
        }
      sl = quarter_sum<real>(sl);
      su = quarter_sum<real>(su);
      scap[k] = sact ? (real)A.s_cap[sidx] : (real)0;
      if (len > 0) {
        smode[k] = 0;
        const real slack = (real)64 * M::proj_tol * fmax((real)1, fabs(scap[k]));
        if (sl > scap[k] + slack) empty_set = true;
        if (eq && su < scap[k] - slack) empty_set = true;
        if (eq && scap[k] >= su) smode[k] = 2;
        else if (scap[k] <= sl) smode[k] = 3;
      }
    }
  }

  const real pd_user = uniform_scalar((real)A.pdiag[b]);
  // prox rows live in equilibrated units z' = s z:  1/2 lf z^2 = 1/2 (lf / s^2) z'^2,  dc max(z) = (dc / s) max(z')
  const real lfb = uniform_scalar(A.lf ? (real)(A.lf[b] / (A.flat_scale * A.flat_scale)) : (real)0);
  const real dcb = uniform_scalar(A.dc ? (real)(A.dc[b] / A.max_scale) : (real)0);
  const real dfl = uniform_scalar(A.dfloor ? (real)(A.dfloor[b] * A.max_scale) : (real)0);
  real tau_max = 0;   // warm start of the demand-charge level
  const real sigma = (real)A.sigma, alpha = (real)A.alpha;
  real rho = (real)A.rho0;
  if (pass > 0) {   // fixed penalty retry_rho * 4^(pass - 1)
    rho = (real)A.retry_rho;
    for (int k = 1; k < pass; ++k) rho *= (real)4;
  }
  real qnorm, pd;
  bool plain_windows = false;   // KS == 1: every period outside the session window has lb = ub = 0 (what aco.py:61-79 builds)
  {
    real f[4];
    f[0] = empty_set ? (real)1 : (real)0;
    f[1] = 0; f[2] = 0; f[3] = 0;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { f[1] = fmax(f[1], fabs(qv[c][r])); f[2] = fmax(f[2], ubv[c][r]); }
    if constexpr (KS == 1) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (!((swm[0] >> (4 * c + tt)) & 1u)) f[3] = fmax(f[3], fmax(fabs(slb[c][tt]), fabs(sub[c][tt])));
    }
    block_max<real, 4>(f, Red, lane, wave, NW);
    plain_windows = KS == 1 && f[3] == (real)0;
    qnorm = uniform_scalar(f[1]);
    // scale-free Tikhonov floor reg_rel * |q|_inf / (max(ub) * T_b), LP-like problems only (effective_pdiag)
    pd = uniform_scalar(effective_pdiag<real>(pd_user, (real)A.reg_rel, qnorm, f[2], A.horizon[b], lfb > (real)0 || dcb > (real)0));
    if (f[0] > 0) {   // a session cannot meet its energy row inside its own bounds
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tt = 16 * c + t, ev = 16 * wave + M::rowof(g, r);
          if (evact[r] && tt < Tm) A.x[((size_t)b * N + ev) * Tm + tt] = 0;
        }
      if (A.y_out && !A.y_for_polish_only)
        for (int k = tid; k < A.Mg * Tm; k += NW * 64) A.y_out[(size_t)b * A.Mg * Tm + k] = 0;
      if (tid == 0) {
        A.status[b] = 4; A.iters[b] = 0;
        A.pri[b] = (double)BIGC; A.dua[b] = (double)BIGC; A.obj[b] = 0;
      }
      break;   // (block-uniform) out of the pass loop: the next problem of the queue
    }
  }

#ifdef ACNQP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  // site-row state, replicated in every wave (identical instruction stream => identical bits)
  real z2[MT][CT][4], y2[MT][CT][4], gx[MT][CT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { z2[m][c][r] = 0; y2[m][c][r] = 0; gx[m][c][r] = 0; }

  rho = uniform_scalar(rho);
  real a = sigma + pd + rho, inv_a = uniform_scalar((real)1 / a), inv_rho = uniform_scalar((real)1 / rho);
  __syncthreads();
  for (int j = tid; j < 16 * MT; j += NW * 64) RowDj[j] = rho / (a + rho * RowLam[j]);
  __syncthreads();

  int status = 2, it = 0, n_adapt = 0, best_it = 0;
  real best_score = BIGC;
  real pri = BIGC, dua = BIGC;
  bool done = false, have_prev = false;
  // duals at the previous residual check (infeasibility certificate), kept in single precision: the certificate asks
  // whether v = y - y_prev is a ray (large, A'v ~ 0, negative support); 2^-24 |y| of rounding cannot fake one
  // (in LDS since round 3: Y1P / Y2P; it is read once per residual check and was what the allocator spilled first)
  const real ptol_scale = M::proj_tol;

  // ---- Anderson acceleration state (block-uniform scalars; vectors in C layout) ----------------------
  constexpr int AMX = AM > 0 ? AM : 1;
  // u and f at the previous event, and the correction c applied then, kept as the float it was applied as (the
  // general-shape kernel's and the C twin's form): g of the previous event is u + c exactly -- half the registers of a
  // stored g.  Tile part, then the site-row part (replicated like z2).
  // f of the previous event only ever enters the difference dF = f - f_prev, itself stored as a float: f_prev is kept
  // as a float too (|dF| >= 1e-3 |f| whenever the column is used -- kAaDrift -- so this costs <= 6e-5 of a column).
  real up1[CT][4];
  float fp1[CT][4], cp1[CT][4];
  real up2[MT][CT][4];
  float fp2[MT][CT][4], cp2[MT][CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      up1[c][r] = 0; fp1[c][r] = 0; cp1[c][r] = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) { up2[m][c][r] = 0; fp2[m][c][r] = 0; cp2[m][c][r] = 0.f; }
    }
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  real* AaRed = sm + L.aared;
  real* AaH = sm + L.aah + (size_t)wave * (AMX * AMX + AMX);   // this wave's copy of (H, b)
  float* HistF1 = reinterpret_cast<float*>(sm + L.hist);        // dF ring: [slot][tid][CT*4], then [slot][lane][MT*CT*4]
  float* HistF2 = HistF1 + (size_t)aa_m * L.hist1;
  float* HistG1 = HistF2 + (size_t)aa_m * L.hist2;              // dG ring, same shape
  float* HistG2 = HistG1 + (size_t)aa_m * L.hist1;
  if constexpr (AM > 0) {
    if (aa_m > 0) {
      for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
      // dead ring slots are read (with a zero coefficient): they must hold finite numbers
      float* ring = HistF1;
      const int nring = 2 * aa_m * (L.hist1 + L.hist2);
      for (int k = tid; k < nring; k += NW * 64) ring[k] = 0.f;
      __syncthreads();
    }
  }

  // ---- projection onto B (energy rows), used by the start and by every iteration; `between` is work that is
  // independent of the water-filling and is placed inside it to share a basic block with the first Newton pass
  auto project_B = [&](const real (&zin)[CT][4], auto&& between) __attribute__((always_inline)) {
    // ---- energy rows: exact water-filling in session layout ---------------------------------------
    // z = clip(zh - m) with g(m) = sum_t clip(zh_t - m) = cap: safeguarded Newton on the piecewise-
    // linear g, warm-started at the previous iteration's m (typically one step + one verifying pass).
      // C layout -> session layout through this wave's private LDS scratch (no workgroup barrier)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[(c * 16 + M::rowof(g, r)) * kXS + t] = zin[c][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      real szh[CT][4], sz[CT][4];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          szh[c][tt] = Xw[(c * 16 + se) * kXS + 4 * sh + tt];
          sz[c][tt] = fmin(fmax(szh[c][tt], slb[c][tt]), sub[c][tt]);
        }
      STAMP(7);   // C -> session transpose
      // (the site-row projection is independent of the water-filling; it sits here so that its VALU work
      //  shares one basic block with the first Newton pass and they hide each other's latency)
      between();
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (k == 0 || k < A.K) {   // block-uniform; slot 0 always exists
          const real cap = scap[k];
          const real tol = ptol_scale * fmax((real)1, fabs(cap));
#if defined(ACNQP_ABL) && ACNQP_ABL == 1
          bool need = false;
#else
          bool need = smode[k] == 0;
#endif
          real m = mu[k];
          real lo = eq ? -BIGC : (real)-1;   // inequality: m >= 0, so (-1, .) brackets m = 0
          real hi = BIGC;
          int guard = 0;
          auto newton_pass = [&]() {
            ++guard;
#ifdef ACNQP_STAMPS
            st_acc[5] += 1000;   // diagnostic build: slot 5 counts water-filling passes (x1000)
#endif

            real gl = 0;
            float nl = 0.f;
            if (plain_windows) {   // block-uniform.  Outside the window lb = ub = 0: the clip is 0 and the period never
                                   // counts as interior, so the sums need no window mask (same bits as the masked form)
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                  const real v = szh[c][tt] - m;
                  gl += fmin(fmax(v, slb[c][tt]), sub[c][tt]);
                  nl += ((v > slb[c][tt]) & (v < sub[c][tt])) ? 1.f : 0.f;
                }
            } else {
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                  const bool inw = (swm[k] >> (4 * c + tt)) & 1u;
                  const real v = szh[c][tt] - m;
                  gl += inw ? fmin(fmax(v, slb[c][tt]), sub[c][tt]) : (real)0;
                  nl += (inw && v > slb[c][tt] && v < sub[c][tt]) ? 1.f : 0.f;
                }
            }
            const real gs = quarter_sum<real>(gl);
            const float nf = quarter_sum<float>(nl);
            const real d = gs - cap;
            // (per-lane logic with bitwise & / |: every `&&` / `||` of per-lane conditions was a divergent region of its own)
            const real big_ = BIGC;
            const bool fin = (fabs(d) <= tol) | (!eq & (m <= (real)0) & (d <= (real)0)) | (guard > ACNQP_GUARD_MAX);
            need = need & !fin;
            const bool dpos = d > 0;
            lo = (need & dpos) ? m : lo;
            hi = (need & !dpos) ? m : hi;
            // flat piece with an open bracket (rare): fetch the true bracket ends so the fallback bisects
            const bool open = need & (nf <= 0.f) & !((lo > -big_) & (hi < big_));
            if (__any(open)) {   // the bracket ends are only ever needed here
              real lo_l = big_, hi_l = -big_;
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                  const bool inw = (swm[k] >> (4 * c + tt)) & 1u;
                  lo_l = inw ? fmin(lo_l, szh[c][tt] - sub[c][tt]) : lo_l;
                  hi_l = inw ? fmax(hi_l, szh[c][tt] - slb[c][tt]) : hi_l;
                }
              const real lo0 = quarter_min<real>(lo_l), hi0 = quarter_max<real>(hi_l);
              lo = open ? fmax(lo, lo0) : lo;
              hi = open ? fmin(hi, hi0) : hi;
            }
            const bool bracketed = (lo > -big_) & (hi < big_);
            const real mid = (real)0.5 * (lo + hi);
            bool newton = nf > 0.f;
            real rc = (real)rcp_small(newton ? nf : 1.f);   // formed for every lane (a select, not a divergent region)
            asm volatile("" : "+v"(rc));
            real c_newton = m + d * rc, c_clamped = fmin(fmax(m + d, lo), hi);   // both candidates, then selects
            asm volatile("" : "+v"(c_newton), "+v"(c_clamped));
            real cand = newton ? c_newton : (bracketed ? mid : c_clamped);
            const bool neg = !eq & (cand < (real)0);                      // inequality: multiplier >= 0
            cand = neg ? (real)0 : cand;
            newton = newton & !neg;
            real a_clamped = fmin(fmax(cand, lo), hi);
            asm volatile("" : "+v"(a_clamped));
            real alt = bracketed ? mid : a_clamped;
            alt = (!eq & (alt < (real)0)) ? (real)0 : alt;
            const bool inside = (cand > lo) & (cand < hi);
            cand = inside ? cand : alt;
            newton = newton & inside;
            // A Newton step that keeps every period of the window on its piece of g (same side of
            // lb / ub before and after) is exact: g is linear between m and cand.  One OR-reduction
            // of a flag word over the session's four lanes replaces the verifying pass.
            unsigned moved = 0;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) {
                const bool inw = (swm[k] >> (4 * c + tt)) & 1u;
                const real v0 = szh[c][tt] - m, v1 = szh[c][tt] - cand;
                const bool ch = ((v0 < sub[c][tt]) != (v1 < sub[c][tt])) | ((v0 > slb[c][tt]) != (v1 > slb[c][tt]));
                moved |= (unsigned)(inw & ch);   // (bitwise: `&&` / `||` became divergent branches)
              }
            { const Pair32 p = swap_u32<16>(moved); moved = p.a | p.b; }
            { const Pair32 q = swap_u32<32>(moved); moved = q.a | q.b; }
            // no representable progress (the residual sits at rounding level, typical in fp32): stop
            need = need & (cand != m);
            m = need ? cand : m;
            need = need & !(newton & (moved == 0u));
          };
          newton_pass();                       // peeled: straight-line with the site-row update above
          while (__any(need)) newton_pass();   // rare: the active set of some session changed
          mu[k] = smode[k] == 0 ? m : mu[k];
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
            {   // branch-free: a lane's window bit and mode select among the candidates (same values)
              const bool inw = (swm[k] >> (4 * c + tt)) & 1u;
              real v0 = fmin(fmax(szh[c][tt] - m, slb[c][tt]), sub[c][tt]);
              asm volatile("" : "+v"(v0));   // formed for every lane: the selects below stay selects
              real val = sz[c][tt];
              val = smode[k] == 3 ? slb[c][tt] : val;
              val = smode[k] == 2 ? sub[c][tt] : val;
              val = smode[k] == 0 ? v0 : val;
              sz[c][tt] = inw ? val : sz[c][tt];
            }
        }
      }
      STAMP(8);   // Newton passes
      // session layout -> C layout
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Xw[(c * 16 + se) * kXS + 4 * sh + tt] = sz[c][tt];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) z1[c][r] = Xw[(c * 16 + M::rowof(g, r)) * kXS + t];
  };

  // ---- start: the schedule that ignores the site rows, z1 = Proj_B(-kStartGain q) (every session served as its
  // cost vector prefers, inside its bounds and energy row), with the multiplier that makes it stationary,
  // y1 = -(q + pd z1); site rows at z2 = G z1 = Q (Ghat z1), y2 = 0.  Exact when no site row binds.
  // Warm start (optional, A.warm_x / A.warm_y): z1 = Proj_B(warm_x), the site-row multipliers y2 = warm_y, and the
  // multipliers of the box / energy set that make the pair stationary, y1 = -(pd z1 + q + G' y2).  (Measured on
  // congested closed loops: -25..-40 % iterations against the cold start; starting from the old y1 instead is worse
  // than cold, because the cost vector q changes with the horizon at every MPC step.)
  {
    const bool warm = pass == 0 && A.warm_x != nullptr && A.warm_y != nullptr;   // block-uniform; a retry pass starts cold
    real zs[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        zs[c][r] = -(real)scalar_const(kStartGain) * qv[c][r];
        if (warm) {
          const int tt = 16 * c + t, ev = 16 * wave + M::rowof(g, r);
          const bool ok = ev < N && tt < Tm;
          zs[c][r] = ok ? (real)A.warm_x[((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0)] : (real)0;
        }
      }
    project_B(zs, []() {});
#pragma unroll
    for (int k = 0; k < KS; ++k) mu[k] = 0;   // the multipliers of this one-off projection are no warm start
    const real* FG0 = static_cast<const real*>(A.fragG) + (size_t)__builtin_amdgcn_readfirstlane(wave) * MT * 2 * 4 * 64;
    const real* FQ0 = static_cast<const real*>(A.fragQ);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      vec4 gty = {0, 0, 0, 0};
      if (warm) {
        const real* RS = static_cast<const real*>(A.rowscale);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * m + M::rowof(g, r), tt = 16 * c + t;
            const int ja = A.rowabi[j];
            const bool ok = ja >= 0 && tt < Tm;
            y2[m][c][r] = ok ? (real)A.warm_y[((size_t)b * A.Mg + (ok ? ja : 0)) * Tm + (ok ? tt : 0)] / RS[j] : (real)0;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * wave + t], y2[m][c][s], gty);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[c][r] = z1[c][r];
        y1[c][r] = -(qv[c][r] + pd * z1[c][r] + gty[r]);
        up1[c][r] = z1[c][r] + y1[c][r] * inv_rho;
        cp1[c][r] = 0.f;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        vec4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(FG0[((m * 2 + 0) * 4 + s) * 64 + lane], z1[c][s], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) Pbuf[(size_t)wave * PS + ((m * CT + c) * 4 + r) * 64 + lane] = acc[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      vec4 g0v[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real g0 = 0;
#pragma unroll
          for (int wv = 0; wv < NW; ++wv) g0 += Pbuf[(size_t)wv * PS + ((m * CT + c) * 4 + r) * 64 + lane];
          g0v[m][r] = g0;
        }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        vec4 zt = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) zt = M::mma(FQ0[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], g0v[mi][s], zt);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z2[mo][c][r] = zt[r]; gx[mo][c][r] = zt[r];
          if (!warm) y2[mo][c][r] = 0;
          up2[mo][c][r] = zt[r] + y2[mo][c][r] * inv_rho;
        }
      }
    }
    if (A.pbuf_single) __syncthreads();   // the start used the one slab the first iteration writes next
  }

#ifndef ACNQP_FRAG_RESIDENT
#define ACNQP_FRAG_RESIDENT 2
#endif
  // A-operand fragments of this wave's OWN products (P_w = Ghat_w r0: level 2; x~ = r0 + Ghat_w' e^: level 1) kept in
  // registers across the loop (MT == 1 only: 8 / 16 registers); the Q fragments of the two site-row products, shared by
  // all waves, sit in LDS (FQs).  With both the loop issues no global load for an MFMA operand at all: lone launch
  // 30.1 -> 28.9 ms (level 2; level 1: 30.0), at 4 spilled registers.  (Round 3 first measured the registers-only form
  // with every fragment resident: -3.7 % for +35 spilled registers -- the LDS copy of Q is what made it affordable.)
  constexpr int kFragRes = MT == 1 ? ACNQP_FRAG_RESIDENT : 0;
#ifndef ACNQP_FRAG_PREFETCH
#define ACNQP_FRAG_PREFETCH 0
#endif
  constexpr bool kFragPre = MT == 1 && kFragRes == 0 && ACNQP_FRAG_PREFETCH != 0;
  real fXr[4], fPr[4];
  if constexpr (kFragRes >= 1) {
    const real* FG0r = static_cast<const real*>(A.fragG) + (size_t)__builtin_amdgcn_readfirstlane(wave) * MT * 2 * 4 * 64;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      fXr[s_] = FG0r[(1 * 4 + s_) * 64 + lane];
      if constexpr (kFragRes >= 2) fPr[s_] = FG0r[(0 * 4 + s_) * 64 + lane];
    }
  }
  while (!done) {
    ++it;
    real* Pw = Pbuf + (size_t)(A.pbuf_single ? 0 : (it & 1)) * NW * PS;
    if (!A.pbuf_single) Xw = Pbuf + (size_t)((it + 1) & 1) * NW * PS + (size_t)wave * PS;   // the idle slab's own region
    // Per-lane constants every predicate of the loop body derives from (row types, session windows, modes) are made
    // opaque once per iteration: the compiler then evaluates `rtype == kRowBox`, `(swm >> k) & 1` ... where they are
    // used (one v_cmp each) instead of hoisting dozens of loop-invariant lane masks into SGPR pairs, which it can
    // only keep by spilling them to VGPR lanes (100-450 SGPR spills per instantiation before this).
#pragma unroll
    for (int k = 0; k < KS; ++k) { asm volatile("" : "+v"(swm[k])); asm volatile("" : "+v"(smode[k])); }
    // an offset the compiler cannot see through keeps the fragment loads inside the loop (the base pointers stay
    // kernel arguments, i.e. provably global memory: global_load, not flat_load)
    unsigned frag_off = (unsigned)__builtin_amdgcn_readfirstlane(wave) * (MT * 2 * 4 * 64), zero_off = 0;
    asm volatile("" : "+s"(frag_off), "+s"(zero_off));
    const real* FG = static_cast<const real*>(A.fragG) + frag_off;
    const real* FQ = static_cast<const real*>(A.fragQ) + zero_off;

    // ---- w^ = Q'(rho z2 - y2) and r0 = sigma x - q + rho z1 - y1;  P_w = Ghat_w r0 --------
    vec4 wh[MT][CT], r0[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) r0[c][r] = sigma * x[c][r] - qv[c][r] + rho * z1[c][r] - y1[c][r];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        vec4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(kFragRes >= 2 ? fPr[s] : FG[((m * 2 + 0) * 4 + s) * 64 + lane], r0[c][s], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[(size_t)wave * PS + ((m * CT + c) * 4 + r) * 64 + lane] = acc[r];
      }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        vec4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = M::mma(MT == 1 ? FQs[(0 * 4 + s) * 64 + lane] : FQ[(((mo * MT + mi) * 2 + 0) * 4 + s) * 64 + lane], rho * z2[mi][c][s] - y2[mi][c][s], acc);
        wh[mo][c] = acc;
      }
    }
    // Experiment switch (-DACNQP_FRAG_PREFETCH=1, off): the A-operand fragments of the two products AFTER the barrier
    // requested before it.  Left to itself the compiler issues each of these eight loads right in front of its MFMA
    // (load, s_waitcnt vmcnt(0), MFMA, eight times in a row); hoisting them costs 16 registers over the barrier and
    // the partial-tile sum, i.e. 10 more spilled registers, and buys 1 % (36.25 -> 35.90 ms): the other wave of the SIMD
    // already covers those waits.
    real fXp[4], fQp[4];
    if constexpr (kFragPre) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { fXp[s] = FG[(1 * 4 + s) * 64 + lane]; fQp[s] = FQ[(1 * 4 + s) * 64 + lane]; }
    }
    STAMP(0);   // r0, P_w, w^ (8 MFMA)
    __syncthreads();   // the one barrier of the iteration: all partial tiles are in LDS
    STAMP(1);   // barrier

    // ---- g0 = sum_w P_w;  e^ = w^ - D (g0 + Lam w^);  h^ = (g0 + Lam e^)/a --------------------
    vec4 eh[MT][CT], hh[MT][CT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real g0 = 0;
#pragma unroll
          for (int wv = 0; wv < NW; ++wv) g0 += Pw[(size_t)wv * PS + ((m * CT + c) * 4 + r) * 64 + lane];
          const real w_ = wh[m][c][r];
          const real lam_ = RowLam[16 * m + M::rowof(g, r)];
          const real e_ = w_ - RowDj[16 * m + M::rowof(g, r)] * (g0 + lam_ * w_);
          eh[m][c][r] = e_;
          hh[m][c][r] = (g0 + lam_ * e_) * inv_a;
        }
    if (A.pbuf_single) __syncthreads();   // every wave has read the one slab before the next iteration overwrites it

    STAMP(2);   // partial-tile sum, e^, h^
    // ---- x~ tile = (r0 + Ghat_w' e^)/a, relaxation, clip ------------------------------------
    real zh[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      vec4 acc = r0[c];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(kFragRes >= 1 ? fXr[s] : (kFragPre ? fXp[s] : FG[((m * 2 + 1) * 4 + s) * 64 + lane]), eh[m][c][s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real xn = acc[r] * inv_a;
        zh[c][r] = alpha * xn + ((real)1 - alpha) * z1[c][r] + y1[c][r] * inv_rho;
        x[c][r] = alpha * xn + ((real)1 - alpha) * x[c][r];
      }
    }
    // ---- site rows: G x~ = Q h^ and the pre-projection point zhr (every wave, redundantly) ----------
    real zhr[MT][CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        vec4 zt = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) zt = M::mma(MT == 1 ? FQs[(1 * 4 + s) * 64 + lane] : (kFragPre ? fQp[s] : FQ[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane]), hh[mi][c][s], zt);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gx[mo][c][r] = alpha * zt[r] + ((real)1 - alpha) * gx[mo][c][r];
          zhr[mo][c][r] = alpha * zt[r] + ((real)1 - alpha) * z2[mo][c][r] + y2[mo][c][r] * inv_rho;
        }
      }
    const bool check = (it % A.check_every == 0) || it >= max_iter_p;

    STAMP(3);   // x~ MFMA, Q h^ MFMA
    // ---- Anderson acceleration event: u = (zh, zhr) is the state of the fixed-point map; every
    // kAaPeriod-th iteration extrapolates it from the last aa_m (dF, dG) column pairs (type II).
    // Control flow is block-uniform; the single extra barrier carries the partial dot products.
    // The arithmetic below is branch-free over all AM ring slots (dead slots hold finite stale data
    // and get a zero coefficient), so the independent dot products / reductions overlap.
    if constexpr (AM > 0) {
      if (aa_m > 0 && it % kAaPeriod == 0) {
        const bool col = aa_have_prev;
        const int slot = aa_head;
        const real w0 = wave == 0 ? (real)1 : (real)0;   // the site-row state is replicated: count it once
        real f1[CT][4], f2[MT][CT][4];
        float cq1[CT][4], cq2[MT][CT][4];          // the new dF column, as stored
        real d[AMX + 2];
        {
          real fa = 0, fb = 0;
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              f1[c][r] = zh[c][r] - up1[c][r];
              fa += f1[c][r] * f1[c][r];
              cq1[c][r] = (float)(f1[c][r] - (real)fp1[c][r]);
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                f2[m][c][r] = zhr[m][c][r] - up2[m][c][r];
                fb += f2[m][c][r] * f2[m][c][r];
                cq2[m][c][r] = (float)(f2[m][c][r] - (real)fp2[m][c][r]);
              }
            }
          d[AMX + 1] = fa + w0 * fb;
        }
        {   // store the column pair in slot `slot` (speculatively: it only counts once marked live)
          float* hf = HistF1 + ((size_t)slot * NW * 64 + tid) * (CT * 4);
          float* hg = HistG1 + ((size_t)slot * NW * 64 + tid) * (CT * 4);
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              hf[c * 4 + r] = cq1[c][r];
              hg[c * 4 + r] = (float)(zh[c][r] - (up1[c][r] + (real)cp1[c][r]));
            }
          if (wave == 0) {
            float* hf2 = HistF2 + ((size_t)slot * 64 + lane) * (MT * CT * 4);
            float* hg2 = HistG2 + ((size_t)slot * 64 + lane) * (MT * CT * 4);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  hf2[(m * CT + c) * 4 + r] = cq2[m][c][r];
                  hg2[(m * CT + c) * 4 + r] = (float)(zhr[m][c][r] - (up2[m][c][r] + (real)cp2[m][c][r]));
                }
          }
        }
        // dF_slot . dF_j for every ring slot j, dF_slot . f
#pragma unroll
        for (int j = 0; j < AMX; ++j) {
          const int jj = j < aa_m ? j : aa_m - 1;
          const float* hj = HistF1 + ((size_t)jj * NW * 64 + tid) * (CT * 4);
          const float* hj2 = HistF2 + ((size_t)jj * 64 + lane) * (MT * CT * 4);
          real a1 = 0, a2 = 0;
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a1 += (real)cq1[c][r] * (real)hj[c * 4 + r];
#pragma unroll
              for (int m = 0; m < MT; ++m) a2 += (real)cq2[m][c][r] * (real)hj2[(m * CT + c) * 4 + r];
            }
          d[j] = a1 + w0 * a2;
        }
        {
          real a1 = 0, a2 = 0;
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a1 += (real)cq1[c][r] * f1[c][r];
#pragma unroll
              for (int m = 0; m < MT; ++m) a2 += (real)cq2[m][c][r] * f2[m][c][r];
            }
          d[AMX] = a1 + w0 * a2;
        }
        // (f, g) of this event replace the previous ones now, so that nothing but zh / zhr crosses the barrier
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            fp1[c][r] = (float)f1[c][r];
#pragma unroll
            for (int m = 0; m < MT; ++m) fp2[m][c][r] = (float)f2[m][c][r];
          }
        // block-wide sums: wave reductions (independent chains), one LDS slot per wave, the event's barrier
#pragma unroll
        for (int j = 0; j < AMX + 2; ++j) d[j] = wave_sum<real>(d[j]);
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < AMX + 2; ++j) AaRed[wave * (AMX + 2) + j] = d[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AMX + 2; ++j) {
          real sw = 0;
#pragma unroll
          for (int wv = 0; wv < NW; ++wv) sw += AaRed[wv * (AMX + 2) + j];
          d[j] = sw;
        }
        STAMP(10);   // event: column, dot products, block reduction
        const real fn = sqrt(d[AMX + 1]);
        bool keep = col;
        if (aa_was && fn > (real)scalar_const(kAaSafe) * fn_prev) {
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
        aa_have_prev = true; fn_prev = uniform_scalar(fn); aa_was = false;
        // no extrapolation while the map drifts (|dF_new| <= kAaDrift |f|): the differences are rounding noise
        real dself = 0;   // |dF_new|^2 (d[slot] without a runtime register index)
#pragma unroll
        for (int j = 0; j < AMX; ++j) dself = j == slot ? d[j] : dself;
        if (aa_cnt > 0 && aa_cool == 0 && !check && dself > (real)scalar_const(kAaDrift * kAaDrift) * d[AMX + 1]) {
          // gamma = (H + eta I)^-1 b: Gauss-Jordan on the augmented AM x (AM + 1) system spread over the
          // wave, lane 8 i + j holding entry (i, j) (H is a regularised Gram matrix: no pivoting); a few
          // registers per lane instead of the whole matrix in every lane.
          static_assert(AMX <= 7, "one 8 x 8 lane tile holds the augmented system");
          const int gi = lane >> 3, gj = lane & 7;
          real tr = 0;
#pragma unroll
          for (int i = 0; i < AMX; ++i) tr += AaH[i * AMX + i];          // dead slots hold zeros
          const real eta = (real)scalar_const(kAaReg) * tr + (real)scalar_const(1e-300);
          real ae = 0;
          if (gi < AMX && gj <= AMX) ae = gj < AMX ? AaH[gi * AMX + gj] : AaH[AMX * AMX + gi];
          if (gi < AMX && gi == gj) ae = ((aa_valid >> gi) & 1u) ? ae + eta : (real)1;
#pragma unroll
          for (int k = 0; k < AMX; ++k) {
            const real piv = lane_value(ae, 9 * k);
            const real rk = __shfl(ae, 8 * k + gj);
            const real ck = __shfl(ae, 8 * gi + k);
            const real rs = rk * rcp_nr(piv);
            ae = gi == k ? rs : ae - ck * rs;
          }
          real gam[AMX];
#pragma unroll
          for (int j = 0; j < AMX; ++j) gam[j] = __shfl(ae, 8 * j + AMX);
          STAMP(11);   // event: LDL' solve
          real cor1[CT][4], cor2[MT][CT][4];
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cor1[c][r] = 0;
#pragma unroll
              for (int m = 0; m < MT; ++m) cor2[m][c][r] = 0;
            }
#pragma unroll
          for (int j = 0; j < AMX; ++j) {
            const int jj = j < aa_m ? j : aa_m - 1;
            const float* hg = HistG1 + ((size_t)jj * NW * 64 + tid) * (CT * 4);
            const float* hg2 = HistG2 + ((size_t)jj * 64 + lane) * (MT * CT * 4);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                cor1[c][r] += gam[j] * (real)hg[c * 4 + r];
#pragma unroll
                for (int m = 0; m < MT; ++m) cor2[m][c][r] += gam[j] * (real)hg2[(m * CT + c) * 4 + r];
              }
          }
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cp1[c][r] = (float)cor1[c][r];            // kept as float; u = g - c with exactly that c
              zh[c][r] -= (real)cp1[c][r];
#pragma unroll
              for (int m = 0; m < MT; ++m) { cp2[m][c][r] = (float)cor2[m][c][r]; zhr[m][c][r] -= (real)cp2[m][c][r]; }
            }
          aa_was = true;
        } else {
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cp1[c][r] = 0.f;
#pragma unroll
              for (int m = 0; m < MT; ++m) cp2[m][c][r] = 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            up1[c][r] = zh[c][r];
#pragma unroll
            for (int m = 0; m < MT; ++m) up2[m][c][r] = zhr[m][c][r];
          }
      }
    }
    STAMP(9);   // Anderson event (amortised)
    project_B(zh, [&]() __attribute__((always_inline)) {
    // ---- site rows: projection of zhr onto C, y2 (every wave, redundantly) ---
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        int rty[4];
        row_types(mo, rty);
        const real pkc = A.peak ? peak_at(c) : BIGC;   // block-uniform branch: only sites with a peak row load it
        // Branch-free over the row types (a lane's type depends on its row, so every `if` here was a divergent region:
        // s_and_saveexec / s_cbranch_execz / s_or exec -- 36 of them per iteration in every wave for four registers):
        // every candidate is formed and the type selects one.  Same values as the branches gave.
        real scl[2], lim4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { lim4[r] = RowLim[16 * mo + M::rowof(g, r)]; asm volatile("" : "+v"(lim4[r])); }   // (not sunk back into a branch)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {   // radial clip factor of the register pairs (0,1), (2,3)
          const real re = zhr[mo][c][2 * pr], im = zhr[mo][c][2 * pr + 1], lim = lim4[2 * pr];
          const real n2 = re * re + im * im;
          const bool clip = rty[2 * pr] == kRowSocRe && n2 > lim * lim;
          const real n2s = clip ? n2 : (real)1;            // rsqrt of a harmless argument where the lane does not clip
          const real f = lim * rsqrt_nr(n2s);
          scl[pr] = clip ? f : (real)1;
        }
        const real quadf = rho / (rho + lfb);
        const real big_s = BIGC;   // once, outside the selects (the opaque constant inside a select made each a branch)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real zh_ = zhr[mo][c][r];
          const int ty = rty[r];
          // a scale factor (SOC pair / quadratic row; 1 otherwise) and an upper limit (box / peak row; +big otherwise):
          // zh * 1 and min(., big) leave the value untouched, so one multiply and one min serve every row type
          // (kRowMax / unused rows: zn = zh_, the horizon-wide prox follows)
          real fac = ((ty == kRowSocRe) | (ty == kRowSocIm)) ? scl[r >> 1] : (real)1;
          fac = ty == kRowQuad ? quadf : fac;
          real cap_ = ty == kRowBox ? lim4[r] : big_s;
          cap_ = ty == kRowPeak ? pkc : cap_;
          const real zn = fmin(zh_ * fac, cap_);
          y2[mo][c][r] = rho * (zh_ - zn);
          z2[mo][c][r] = zn;
        }
      }
    }
    // ---- demand charge: prox of dc * max(max_t z_t, floor) over the whole horizon of the "max" row:
    // z_t = min(zh_t, tau), tau = max(floor, root of sum_t (zh_t - tau)+ = dc / rho) (Newton on a convex
    // piecewise-linear function; the row's 16 periods are the 16 lanes of a DPP row).
    if (A.dc != nullptr && dcb > (real)0) {   // block-uniform
      real zv[CT];
      bool mine = false;
#pragma unroll
      for (int c = 0; c < CT; ++c) zv[c] = 0;
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        int rty[4];
        row_types(mo, rty);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rty[r] == kRowMax) {
            mine = true;
#pragma unroll
            for (int c = 0; c < CT; ++c) zv[c] = z2[mo][c][r];   // = zh of that row (left unprojected above)
          }
      }
      const real cw = dcb * inv_rho;
      real vmax_l = -BIGC;
#pragma unroll
      for (int c = 0; c < CT; ++c) vmax_l = (16 * c + t < Tm) ? fmax(vmax_l, zv[c]) : vmax_l;
      const real vmax = row_max<real>(vmax_l);
      real tau = tau_max;
      bool need = mine;
      int guard = 0;
      while (__any(need)) {
        ++guard;
        real sl = 0;
        float nl = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const real dd = zv[c] - tau;
          const bool on = (16 * c + t < Tm) && dd > (real)0;
          sl += on ? dd : (real)0;
          nl += on ? 1.f : 0.f;
        }
        const real S = row_sum<real>(sl);
        const float nn = row_sum<float>(nl);
        const real f = S - cw;
        const real tn = nn > 0.f ? tau + f * (real)rcp_small(nn) : vmax - cw;
        const bool fin = fabs(f) <= M::proj_tol * fmax((real)1, cw) * (real)16 || tn == tau || guard > 60;
        tau = (need && !fin) ? tn : tau;
        need = need && !fin;
      }
      tau_max = tau;
      const real lev = fmax(tau, dfl);
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        int rty[4];
        row_types(mo, rty);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rty[r] == kRowMax) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              const real zh_ = z2[mo][c][r];
              const real zn = (16 * c + t < Tm) ? fmin(zh_, lev) : zh_;
              y2[mo][c][r] = rho * (zh_ - zn);
              z2[mo][c][r] = zn;
            }
          }
      }
    }
    });
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) y1[c][r] = rho * (zh[c][r] - z1[c][r]);

    STAMP(4);   // session -> C transpose + y1
    STAMP(5);   // site rows
    // ---- residuals, termination, rho adaptation (block-uniform decisions) -----------------
    if (check) {
      real v[6];   // pri, dua, |Ax| |z|, -, |Px|, |A'y|
      v[0] = v[1] = v[2] = v[3] = v[4] = v[5] = 0;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        vec4 gty = {0, 0, 0, 0};   // (G' y2) tile of this wave
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * wave + t], y2[m][c][s], gty);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[0] = fmax(v[0], fabs(x[c][r] - z1[c][r]));
          v[1] = fmax(v[1], fabs(pd * x[c][r] + qv[c][r] + y1[c][r] + gty[r]));
          v[2] = fmax(v[2], fmax(fabs(x[c][r]), fabs(z1[c][r])));
          v[4] = fmax(v[4], fabs(pd * x[c][r]));
          v[5] = fmax(v[5], fabs(y1[c][r] + gty[r]));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[0] = fmax(v[0], fabs(gx[m][c][r] - z2[m][c][r]));
            v[2] = fmax(v[2], fmax(fabs(gx[m][c][r]), fabs(z2[m][c][r])));
          }
      }
      block_max<real, 6>(v, Red, lane, wave, NW);
      pri = uniform_scalar(v[0]);
      dua = uniform_scalar(v[1]);
      const real npri = v[2];
      const real ndua = fmax(fmax(v[4], v[5]), qnorm);
      const real eps_p = (real)A.eps_abs + (real)A.eps_rel * npri;
      const real eps_d = (real)A.eps_abs + (real)A.eps_rel * ndua;
      if (pri <= eps_p && dua <= eps_d) { status = 1; done = true; }
      if (!done && have_prev) {
        // ---- primal infeasibility certificate (OSQP's, generalised to the sets B and C) -----------
        // v = y - y(previous check).  If A'v ~ 0 and the support function of B x C at v is negative,
        // no point of B x C can satisfy A r = z: infeasible.  For B the support function of a session
        // is bounded above by phi(l) = l cap + sum_t [ub (v_t - l)+ + lb (v_t - l)-] for any admissible l.
        real w6[3];
        w6[0] = w6[1] = w6[2] = 0;   // |v|, |v1 + G'v2|, "unbounded direction" flag
        real ssum = 0;               // support-function bound, summed over the block
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          vec4 gtv = {0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              gtv = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * wave + t], y2[m][c][s] - (real)Y2P[(m * CT + c) * 4 + (s)], gtv);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const real v1 = y1[c][r] - (real)Y1P[c * 4 + r];
            w6[0] = fmax(w6[0], fabs(v1));
            w6[1] = fmax(w6[1], fabs(v1 + gtv[r]));
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const real v2 = y2[m][c][r] - (real)Y2P[(m * CT + c) * 4 + (r)];
              w6[0] = fmax(w6[0], fabs(v2));
            }
          }
        }
        block_max<real, 3>(w6, Red, lane, wave, NW);
        const real vn = w6[0];
        const real vtol = (real)scalar_const(1e-4) * vn;
        if (vn > (real)scalar_const(1e-12) * fmax((real)1, qnorm) && w6[1] <= vtol) {
          real bad = 0;
          if (wave == 0) {   // the site-row state is replicated: count it once
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                int rty[4];
                row_types(m, rty);
                const real pkc = peak_at(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const real v2 = y2[m][c][r] - (real)Y2P[(m * CT + c) * 4 + (r)];
                  const int ty = rty[r];
                  if (ty == kRowBox) { ssum += RowLim[16 * m + M::rowof(g, r)] * fmax(v2, (real)0); if (v2 < -vtol) bad = 1; }
                  else if (ty == kRowPeak) {
                    if (pkc < BIGC) ssum += pkc * fmax(v2, (real)0); else if (v2 > vtol) bad = 1;
                    if (v2 < -vtol) bad = 1;
                  } else if (ty == kRowSocRe) {
                    const real vi = y2[m][c][(r + 1) & 3] - (real)Y2P[(m * CT + c) * 4 + ((r + 1) & 3)];
                    ssum += RowLim[16 * m + M::rowof(g, r)] * sqrt(v2 * v2 + vi * vi);
                  } else if (ty == kRowSocIm) {
                  } else if (fabs(v2) > vtol) bad = 1;   // free / quadratic rows admit no ray
                }
              }
          }
          // sessions: transpose v1 to session layout and bound each session's support function
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xw[(c * 16 + M::rowof(g, r)) * kXS + t] = y1[c][r] - (real)Y1P[c * 4 + r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          real sv[CT][4];
          unsigned covered = 0;
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) sv[c][tt] = Xw[(c * 16 + se) * kXS + 4 * sh + tt];
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            if (k == 0 || k < A.K) {
              covered |= swm[k];
              real lmin_l = BIGC, lmax_l = -BIGC;
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                  if ((swm[k] >> (4 * c + tt)) & 1u) { lmin_l = fmin(lmin_l, sv[c][tt]); lmax_l = fmax(lmax_l, sv[c][tt]); }
              real lam3[3];
              lam3[0] = quarter_min<real>(lmin_l);
              lam3[1] = quarter_max<real>(lmax_l);
              lam3[2] = 0;
              real best = BIGC;
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                real l_ = lam3[j];
                if (!eq) l_ = fmax(l_, (real)0);
                real ph = 0;
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                  for (int tt = 0; tt < 4; ++tt)
                    if ((swm[k] >> (4 * c + tt)) & 1u) {
                      const real dv = sv[c][tt] - l_;
                      ph += sub[c][tt] * fmax(dv, (real)0) + slb[c][tt] * fmin(dv, (real)0);
                    }
                ph = quarter_sum<real>(ph) + l_ * scap[k];
                best = fmin(best, ph);
              }
              if (smode[k] != 4 && sh == 0) ssum += best;   // one lane per session
            }
          }
          // periods outside every window are pinned to lb = ub (= 0): support lb * v
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
              if (!((covered >> (4 * c + tt)) & 1u)) ssum += slb[c][tt] * sv[c][tt];
          __builtin_amdgcn_wave_barrier();
          real tot = wave_sum<real>(ssum);
          real bd[1] = {bad};
          __syncthreads();
          if (lane == 0) Red[wave * kNumRed + 7] = tot;
          block_max<real, 1>(bd, Red, lane, wave, NW);
          real stot = 0;
          for (int wv = 0; wv < NW; ++wv) stot += Red[wv * kNumRed + 7];
          __syncthreads();
          if (bd[0] == (real)0 && stot < -vtol) { status = 3; done = true; }
        }
      }
      if (!done) {   // snapshot for the next certificate test
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) Y1P[c * 4 + r] = (float)y1[c][r];
        if (wave == 0) {   // the site-row state is replicated: one copy; every other wave's reads of the old one lie before
                           // the barriers of this check's reductions
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) Y2P[(m * CT + c) * 4 + r] = (float)y2[m][c][r];
        }
        have_prev = true;
      }
      const real tiny_ = (real)scalar_const(1e-300);
      const real score = fmax(pri / fmax(eps_p, tiny_), dua / fmax(eps_d, tiny_));
      if (score < (real)scalar_const(kStallGain) * best_score) { best_score = uniform_scalar(score); best_it = it; }
      const bool inacc = inaccurate_ok<real>(pri, dua, npri, ndua, A.eps_abs, A.eps_rel, A.inacc_floor);
      const bool stalled = A.stall_iters > 0 && it - best_it >= A.stall_iters && score <= (real)scalar_const(kStallNear) * best_score;
      bool hand_over = false;
      if (!done && pass == 0 && A.polish_iters > 0 && it >= A.polish_iters) {   // block-uniform
        // rows the polish's Schur system would have: one per tight box / peak row, two per tight disc (normal + tangent)
        real cnt = 0;
        const real ytol = (real)scalar_const(1e-9) * fmax((real)1, qnorm);
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int mo = 0; mo < MT; ++mo) {
            int rty[4];
            row_types(mo, rty);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const real yr = y2[mo][c][r], yi = y2[mo][c][(r + 1) & 3];
              const bool disc = rty[r] == kRowSocRe;
              const real mag = disc ? sqrt(yr * yr + yi * yi) : yr;
              const bool counts = (disc | (rty[r] == kRowBox) | (rty[r] == kRowPeak)) & (16 * c + t < Tm) & (mag > ytol);
              cnt += counts ? (disc ? (real)2 : (real)1) : (real)0;
            }
          }
        cnt = wave_sum<real>(cnt);   // the site-row state is replicated: every wave counts the same rows
        hand_over = cnt + (real)8 <= (real)A.pol_rows;
      }
      if (done) {
      } else if (hand_over) {
        status = kStatusPolish;   // not converged after polish_iters iterations: the polish kernel takes over from (z1, y2)
        done = true;
      } else if (it >= max_iter_p || stalled) {
        done = true;
        if (inacc) status = 5;   // solved, inaccurately
      }
      else if (adapt_p > 0 && it % adapt_p == 0) {
        const real e12_ = (real)scalar_const(1e-12);
        const real sp = pri / fmax(npri, e12_);
        const real sd = dua / fmax(ndua, e12_);
        const real ratio = sqrt(sp / fmax(sd, (real)scalar_const(1e-30)));
        const real tol_eff = (real)A.adapt_tol * ((real)1 + (real)n_adapt * (real)(1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < (real)1 / tol_eff) {
          ++n_adapt;
          rho = uniform_scalar(fmin(fmax(rho * ratio, (real)scalar_const(1e-6)), (real)scalar_const(1e6)));
          a = sigma + pd + rho;
          inv_a = uniform_scalar((real)1 / a);
          inv_rho = uniform_scalar((real)1 / rho);
          for (int j = tid; j < 16 * MT; j += NW * 64) RowDj[j] = rho / (a + rho * RowLam[j]);
          __syncthreads();   // block-uniform branch: every wave sees the new D before the next iteration
          if constexpr (AM > 0) {
            if (aa_m > 0) {   // the fixed-point map changed: restart the ring from the current (z, y)
              aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
              __builtin_amdgcn_wave_barrier();
              for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  up1[c][r] = z1[c][r] + y1[c][r] / rho;
                  cp1[c][r] = 0.f;
#pragma unroll
                  for (int m = 0; m < MT; ++m) { up2[m][c][r] = z2[m][c][r] + y2[m][c][r] / rho; cp2[m][c][r] = 0.f; }
                }
            }
          }
        }
      }
    }
    STAMP(6);   // residual check (amortised)
  }
  it_total += it;
  // ---- results of this pass: the feasible iterate z1 is the schedule (kept if it beats the earlier passes) -----
  if (pass == 0 || status_rank(status) > status_rank(best_status)) {   // block-uniform
  best_status = status;
  real ol = 0;
  int lane_o = lane;   // opaque copy: the store predicates are evaluated here, not kept alive across the loop
  asm volatile("" : "+v"(lane_o));
  const int g_o = lane_o >> 4, t_o = lane_o & 15;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tt = 16 * c + t_o, ev = 16 * wave + M::rowof(g_o, r);
      if (ev < N && tt < Tm) {
        A.x[((size_t)b * N + ev) * Tm + tt] = (double)z1[c][r];
        ol += ((real)0.5 * pd_user * z1[c][r] + qv[c][r]) * z1[c][r];
      }
    }
  if (A.y_out && wave == 0 && (!A.y_for_polish_only || status == kStatusPolish)) {   // site-row multipliers in the caller's row order and units (the state is replicated: wave 0 writes)
    const real* RS = static_cast<const real*>(A.rowscale);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * m + M::rowof(g_o, r), tt = 16 * c + t_o;
          const int ja = A.rowabi[j];
          if (ja >= 0 && tt < Tm) A.y_out[((size_t)b * A.Mg + ja) * Tm + tt] = (double)(y2[m][c][r] * RS[j]);
        }
  }
  ol = wave_sum<real>(ol);
  __syncthreads();
  if (lane == 0) Red[wave * kNumRed] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < NW; ++wv) o += Red[wv * kNumRed];
    A.status[b] = status;
    A.pri[b] = (double)pri;
    A.dua[b] = (double)dua;
    A.obj[b] = (double)o;
    if (status == kStatusPolish) A.pol_list[atomicAdd(A.pol_count, 1)] = b;
  }
  __syncthreads();   // Red is reused by the next pass
  }
  if (tid == 0) A.iters[b] = it_total;
#ifdef ACNQP_STAMPS
  if (lane == 0 && b < 1024)
    for (int k = 0; k < 12; ++k) g_stamps[(b * 16 + wave) * 12 + k] = st_acc[k];
#endif
  if (!retry_wanted(pass, A.retry_passes, status, it, A.stall_iters, A.adapt_every)) break;
  }   // passes
  }   // work queue
#undef BIGC
}

}  // namespace acnqp
