// Large-site kernel of the batched MPC QP solver for gfx950: N > 64 EVSEs (up to 1024), horizon <= 48, <= 48 site
// rows -- BASELINE.json configs[4] (512 EVSE x 48 periods, load_flattening) and the reference's big single instances
// (aco.py:403-408, adacharge.py:249-276) when they fit that horizon.
//
// Same ADMM as acn_qp_tiled.hpp (see there), but an N x T iterate no longer fits the registers / LDS of one
// workgroup (512 x 48 doubles = 192 KB per array), so the iterates x, z1, y1 and the constants q, lb, ub STREAM
// through HBM once per iteration, in MFMA fragment order ([EVSE tile][column tile][register][lane]: every wave
// load is one contiguous 512-byte row), and the two cross-EVSE products are real GEMMs on the matrix cores:
//
//   x~ tile   = (r0 + Ghat[:, tile]' e^) / a      N x Mr x T   (Mr = padded site rows)      v_mfma_f64_16x16x4_f64
//   P         = sum over tiles Ghat[:, tile] r0   Mr x N x T
//
// One workgroup of 8 waves per problem; wave w owns EVSE tiles w, w + 8, ...  Per iteration ONE fused pass over the
// tiles: load (x, z1, y1, q, lb, ub) -> r0 -> x~ (MFMA, e^ from LDS) -> relaxation -> box + energy-row projection
// (water-filling along the 16-lane DPP row that holds one EVSE's periods, all column tiles in registers) -> y1 ->
// store (x, z1, y1) -> the NEW r0 of the tile.  Nine N x T arrays cross HBM per iteration (9 * 8 * NP * TP bytes,
// non-temporal: they must not evict the shared site fragments from L2): the kernel is HBM-bound by design.
// P = Ghat r0 is formed in ROUNDS of 8 tiles: every wave parks its tile's new r0 in an LDS slab (it is the MFMA B
// operand as it stands), and the wave that OWNS an output tile (m, c) of P accumulates the 8 staged tiles into its
// one or two accumulators -- 8-16 registers instead of the 72 a per-wave partial P would pin, no cross-wave
// reduction, and a fixed summation order (deterministic).  The <= 48 x 48 site-row state (z2, y2, G x, e^, h^)
// lives in LDS and is advanced by the same owner waves.
//
// Anderson acceleration (type II, the other kernels' rule: an event every fifth iteration, ring and u / f / g in the
// workspace): an event iteration splits the fused pass in two -- pass A computes every tile's pre-projection point,
// stores it and accumulates the event's dot products; after the block-wide reduction and the small solve, pass B
// extrapolates the stored points and runs the projection / y1 / new r0 half of the fused pass (eleven more array passes
// per event, i.e. +25 % bytes per iteration on average, for 2-3x fewer iterations on congested problems).  The primal
// infeasibility certificate and the demand-charge row (one wave, after the site tiles) are the long-horizon kernel's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acn_qp_tiled.hpp"

namespace acnqp {

typedef unsigned ws_v2u_s __attribute__((ext_vector_type(2)));
typedef unsigned ws_v4u_s __attribute__((ext_vector_type(4)));
typedef double ws_d2_s __attribute__((ext_vector_type(2)));
typedef float ws_f2_s __attribute__((ext_vector_type(2)));

// Waves per workgroup is a template parameter NWV: 4 (two workgroups = two problems share a CU and fill each other's
// memory phases: best throughput on large batches) or 8 (one workgroup per CU: a single problem finishes sooner).
// The summation order of every reduction is the same for both, so the choice never changes a result bit.

struct StreamArgs {
  TiledArgs t;          // same site / problem / result / option fields as the tiled kernel (fragG covers NP / 16 tiles)
  double* work;         // [B][ws_per_problem]
  long long ws_per_problem;
};

constexpr int kStreamAccelMax = 5;   // Anderson columns (the ring lives in the workspace)

// doubles of workspace one problem needs (accel = Anderson columns in use)
// Compressed bounds of an EVSE tile (`flat tiles` in the kernel; the long-horizon kernel's flat items, acn_qp_long.hpp):
// per register row and lane one (l, u) pair of doubles, and per lane the 4 x CT period bits of its four rows
#ifndef ACNQP_STREAM_FLAT_BOUNDS
#define ACNQP_STREAM_FLAT_BOUNDS 1
#endif
constexpr int kStreamFlatTiles = 64;   // tiles whose flags fit the LDS table: sites of up to 1,024 EVSEs
__host__ __device__ inline long long stream_flat_doubles(int NP) { return (long long)(NP / 16) * 544; }

__host__ __device__ inline long long stream_workspace(int NP, int CT, int K, int MT, int accel) {
  const long long NT = (long long)(NP / 16) * CT * 256, MS = (long long)MT * CT * 256, DU = NT + MS;
  long long w = 6 * NT + (long long)K * NP + 3 * MS + (DU + 1) / 2 + 64;   // + the certificate's dual snapshot (floats)
  if (accel > 0) w += DU + 3 * DU + (2LL * accel * DU * 4 + 7) / 8;         // zh / zhr of the event; u, f, g; the float rings
  w += stream_flat_doubles(NP);   // at the END of the workspace
  return w;
}

// LDS carve-up (doubles)
struct StreamLds {
  int red, g0h, we, scal, total;
  __host__ __device__ StreamLds(int MT, int CT, int NWV) {
    int o = 0;
    red = o;  o += 2 * NWV * CT * 256;        // the rounds' staged tiles, double-buffered (one barrier per round)
    g0h = red + NWV * CT * 256;               // Ghat z1 (start) / h^: ALIASES the second slab (MT <= NWV) -- h^ lives from
                                              // the eigen step to the site-row update, the slabs only inside the tile passes
    we = o;   o += MT * CT * 256;             // e^: B operand of every tile's x~ product
    scal = o; o += NWV * 8 + 8;
    total = o;
  }
};

#ifndef ACNQP_STREAM_FRONT_FENCE
#define ACNQP_STREAM_FRONT_FENCE 0
#endif

template <int NV, int NWV>
__device__ inline void stream_block_max(double (&v)[NV], double* S, int lane, int wave) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double m = wave_max<double>(v[k]);
    if (lane == 0) S[wave * 8 + k] = m;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double m = S[k];
    for (int wv = 1; wv < NWV; ++wv) m = fmax(m, S[wv * 8 + k]);
    v[k] = m;
  }
  __syncthreads();
}

template <int CT, int MT, int NWV>
#ifndef ACNQP_STREAM_OCC
#define ACNQP_STREAM_OCC 2
#endif
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? ACNQP_STREAM_OCC : 1) void admm_stream_kernel(const StreamArgs SA_kernarg) {
  constexpr int kStreamWaves = NWV;
  using M = Mfma<double>;
  using vec4 = M::vec4;
  typedef double real;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* sm = reinterpret_cast<real*>(smem_raw);
  const StreamLds L(MT, CT, NWV);
  constexpr int AMX = kStreamAccelMax;
  __shared__ real AaRedS[NWV * (AMX + 2)];               // per-wave partial dot products of an Anderson event
  __shared__ real AaHS[NWV * (AMX * AMX + AMX)];         // every wave's own copy of the Gram matrix and rhs
  __shared__ real ResS[NWV * 8];                          // per wave: the residual terms of a check iteration
  // Clearing a wave's copy: EVERY lane stores (the lanes beyond the 30 entries repeat the last one) -- no divergent
  // region.  Behind the `if (lane < 30)` form the compiler (ROCm 7.2) placed the spill of two values that live across
  // the solver loop BEFORE the exec restore of the region's end: stored with no lane enabled, reloaded as garbage (the
  // reported objective lost its pdiag term; adacharge_amd/store_hazard.py scan_exec_spills now refuses such code).
  static_assert(AMX * AMX + AMX <= 64, "one store per lane clears the Gram matrix and its right-hand side");
  auto aa_clear_idx = []() -> int { const int l = (int)(threadIdx.x & 63); return l < AMX * AMX + AMX ? l : AMX * AMX + AMX - 1; };
  real* RED0 = sm + L.red;
  real* G0H = sm + L.g0h;
  real* WE = sm + L.we;
  real* SC = sm + L.scal;

  // passes: pass 0 as the options state it, then cold fixed-penalty retries of a stalled problem (retry_wanted,
  // acn_qp_tiled.hpp).  The WHOLE body is the pass, with the thread / block ids opaque and the argument block read
  // through a per-pass opaque pointer to the kernarg segment: nothing of a pass is invariant across passes, so no
  // pass-invariant address, predicate or argument is kept alive across the solver loop.
  __shared__ int q_slot;
  __shared__ unsigned char TileFlat[kStreamFlatTiles];   // per EVSE tile: its bounds compress (flat tiles, tile_back)
  for (int q_round = 0;; ++q_round) {   // work queue: this workgroup's next problem (queue_next, acn_qp_tiled.hpp)
  const int q_pos = queue_next(SA_kernarg.t.queue, queue_length(SA_kernarg.t), q_round, &q_slot);
  if (q_pos < 0) break;
  int it_total = 0, best_status = 0;
  for (int pass = 0;; ++pass) {
  typedef const __attribute__((address_space(4))) StreamArgs* KernargP;
  KernargP SAp = (KernargP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(SAp));
  const auto& SA = *SAp;
  const auto& A = SA.t;
  int b_ = q_pos, tid = threadIdx.x;
  asm volatile("" : "+v"(b_));
  asm volatile("" : "+v"(tid));
  const int wg_ = __builtin_amdgcn_readfirstlane(b_);
  const int b = __builtin_amdgcn_readfirstlane(A.order ? A.order[wg_] : wg_);   // the problem this workgroup solves (a uniform value: the load alone would make it a vector register)
  const int max_iter_p = pass == 0 ? A.max_iter : min(A.max_iter, A.retry_max_iter);
  const int adapt_p = pass == 0 ? A.adapt_every : 0;
  // wave index as a scalar: tile bases become SGPR addresses (global_load saddr + lane offset) instead of one 64-bit
  // VGPR address per array, column tile and register
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // lane, g, t are re-derived from an opaque copy at the top of every tile iteration (RELANE): per-lane addresses
  // (session table, multipliers, site-matrix columns) are then computed where they are used instead of being hoisted
  // out of the solver loop and held -- or spilled -- across it
  int lane = tid & 63;
  int g = lane >> 4, t = lane & 15;
#define RELANE() do { asm volatile("" : "+v"(lane)); g = lane >> 4; t = lane & 15; } while (0)
  const int N = A.N, Tm = A.Tm, NP = A.NP, K = A.K;
  const int NE = NP >> 4;                 // EVSE tiles
  const long long NT = (long long)NE * CT * 256;
  // the workspace belongs to the workgroup SLOT, not to the problem, when the launch runs off the queue (every pass
  // initialises what it reads): grid x ws_per_problem instead of B x ws_per_problem
  // (the slot index is made opaque per pass like the ids above: as a plain blockIdx.x every workspace address was
  //  invariant across the passes, was hoisted out of the pass loop and lived -- spilled -- across the solver loop:
  //  <3,3,4> 90 -> 188 spilled registers, configs[4] leg 423 -> 460 ms)
  int ws_slot_ = (int)blockIdx.x;
  asm volatile("" : "+v"(ws_slot_));
  ws_slot_ = __builtin_amdgcn_readfirstlane(ws_slot_);
  real* W0 = SA.work + (size_t)(A.ws_by_slot ? ws_slot_ : b) * SA.ws_per_problem;
  real *Xs = W0, *Z1s = Xs + NT, *Y1s = Z1s + NT, *Qs = Y1s + NT, *LBs = Qs + NT, *UBs = LBs + NT;
  real* MU = UBs + NT;                    // [K][NP]
  // site-row state (z2, y2, G x) in tile-fragment order: touched by the owner waves once per iteration, L2-resident
  real* Z2 = MU + (size_t)K * NP;
  real* Y2 = Z2 + MT * CT * 256;
  real* GX = Y2 + MT * CT * 256;
  const long long MS = (long long)MT * CT * 256, DU = NT + MS;
  const int aa_m = min(A.accel_mem, kStreamAccelMax);
  // The certificate's snapshot and the Anderson state live behind GX; their pointers are formed where they are used
  // (residual check, event, restart) from an opaque copy of the workspace base, so that none of them is a live
  // register pair across the solver loop:
  //   Y1P [NT] Y2P [MS] floats: duals at the previous residual check (certificate)
  //   ZHs [NT + MS]: the event's pre-projection points; UP, FP, GP [NT + MS] each: u, f, g at the previous event
  //   HF, HG [aa_m][NT + MS] floats: the dF / dG rings.  Index uo = tile index, the site part after the NT tile entries.
  const long long off_snap = 6 * NT + (long long)K * NP + 3 * MS, off_aa = off_snap + (DU + 1) / 2;
#define ACNQP_STREAM_AA_PTRS()                                                                       \
  unsigned aoff_ = 0;                                                                                \
  asm volatile("" : "+s"(aoff_));                                                                    \
  real* ZHs = W0 + off_aa + aoff_;                                                                   \
  real* UP = ZHs + DU;                                                                               \
  real* FP = UP + DU;                                                                                \
  real* GP = FP + DU;                                                                                \
  float* HF = reinterpret_cast<float*>(GP + DU);                                                     \
  float* HG = HF + (size_t)aa_m * DU;                                                                \
  (void)ZHs; (void)UP; (void)FP; (void)GP; (void)HF; (void)HG
  const real* FG = static_cast<const real*>(A.fragG);
  const real* FQ = static_cast<const real*>(A.fragQ);
  const real* Gm = static_cast<const real*>(A.G);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);
  const bool eq = A.s_eq[b] != 0;
  const real sigma = A.sigma, alpha = A.alpha;
  const real lfb = A.lf ? A.lf[b] / (A.flat_scale * A.flat_scale) : 0.0;
  // demand charge (acn_qp_tiled.hpp): weight, floor and the padded index of the "max" row (uniform)
  const real dcb = A.dc ? A.dc[b] / A.max_scale : 0.0;
  const real dfl = A.dfloor ? A.dfloor[b] * A.max_scale : 0.0;
  int jdc = -1;
  if (dcb > 0.0)
    for (int j = 0; j < 16 * MT; ++j) jdc = A.rowtype[j] == kRowMax ? j : jdc;
  const bool dc_on = jdc >= 0;

  // Layout of the six state arrays (x, z1, y1, q, lb, ub): 16 x 16 tiles in MFMA fragment order with the four
  // accumulator registers of a lane stored as two adjacent PAIRS (the long-horizon kernel's layout) -- element
  // (tile, r, lane) at tile * 256 + (r / 2) * 128 + lane * 2 + r % 2 -- so that one 16-byte access per lane moves two
  // registers: half the memory instructions and 1 KB contiguous per wave access instead of 512 B.  The Anderson arrays
  // (zh of the event, u, f, g, the float rings) use the same layout; the site-row state keeps the plain order [tile][r][lane].
  auto fidx = [&](int e, int c, int r) -> size_t { return (size_t)(e * CT + c) * 256 + (size_t)((r >> 1) * 128 + lane * 2 + (r & 1)); };
  // The hot passes (tile_front / tile_back, the Anderson event's passes) address the workspace through ONE buffer
  // descriptor in scalar registers: element (array, uniform index u)[lane] = descriptor + (array offset + u) * 8 as the
  // scalar offset + lane * 8 as the one 32-bit lane offset every access shares.  With plain pointers the compiler kept
  // a 64-bit per-lane address for every (array, column tile) -- 24 registers in the fused pass, recomputed per tile --
  // and spilled around them (416 bytes of scratch traffic per lane and tile against 864 useful, round 3).  The cold
  // paths (start, restart, residual check, certificate, output) keep the pointers; both name the same memory.
  const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(W0, 0, (int)(SA.ws_per_problem * 8), 0x00020000);
  const bool flat_on = ACNQP_STREAM_FLAT_BOUNDS && NE <= kStreamFlatTiles;
  const unsigned cblu = (unsigned)(SA.ws_per_problem - stream_flat_doubles(NP)) * 8u;   // [tile][row][lane] (l, u)
  const unsigned cbm = cblu + (unsigned)NE * 4096u;                                     // [tile][lane] period bits
  constexpr int kNt = 2;   // gfx940+ cache policy bit 1 = nt: what __builtin_nontemporal_load / _store emit
  const unsigned oX = 0, oZ1 = (unsigned)NT, oY1 = 2 * (unsigned)NT, oQ = 3 * (unsigned)NT, oLB = 4 * (unsigned)NT, oUB = 5 * (unsigned)NT;
  auto fidp = [&](int e, int c, int rp) -> unsigned { return (unsigned)((e * CT + c) * 256 + rp * 128); };   // registers 2 rp, 2 rp + 1
  auto ld2_nt = [&](unsigned arr, unsigned u, real& a, real& b) __attribute__((always_inline)) {
    const ws_d2_s d = __builtin_bit_cast(ws_d2_s, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, (arr + u) * 8u, kNt));
    a = d.x; b = d.y;
  };
  auto st2_nt = [&](unsigned arr, unsigned u, real a, real b) __attribute__((always_inline)) {
    const ws_d2_s d = {a, b};
    const ws_v4u_s q = __builtin_bit_cast(ws_v4u_s, d);
    __builtin_amdgcn_raw_buffer_store_b128(q, wsr, (unsigned)lane * 16u, (arr + u) * 8u, kNt);
    // gfx950 store-data hazard (acn_qp_long.hpp st2): three wait states with the very register
    // tuple the store reads kept live (the asm takes the 128-bit operand itself, not the doubles it was built from)
    asm volatile("s_nop 2" ::"v"(q));
  };
  // the Anderson arrays (pre-projection points of the event, u, f, g: doubles; the dF / dG rings: floats) take the same
  // pair layout over DU = NT + MS elements (EVSE tiles, then the site tiles): tile index `tix`, pair rp at tix * 256 + rp * 128
  const unsigned oZH = (unsigned)off_aa, oUP = oZH + (unsigned)DU, oFP = oUP + (unsigned)DU, oGP = oFP + (unsigned)DU;
  const unsigned bHF = (oGP + (unsigned)DU) * 8u, bHG = bHF + (unsigned)aa_m * (unsigned)DU * 4u;   // byte offsets
  auto ld2 = [&](unsigned arr, unsigned u, real& a, real& b) __attribute__((always_inline)) {
    const ws_d2_s d = __builtin_bit_cast(ws_d2_s, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, (arr + u) * 8u, 0));
    a = d.x; b = d.y;
  };
  auto st2 = [&](unsigned arr, unsigned u, real a, real b) __attribute__((always_inline)) {
    const ws_d2_s d = {a, b};
    const ws_v4u_s q = __builtin_bit_cast(ws_v4u_s, d);
    __builtin_amdgcn_raw_buffer_store_b128(q, wsr, (unsigned)lane * 16u, (arr + u) * 8u, 0);
    asm volatile("s_nop 2" ::"v"(q));   // store-data hazard, as in st2_nt
  };
  auto ld2f = [&](unsigned boff, unsigned u, float& a, float& b) __attribute__((always_inline)) {
    const ws_f2_s d = __builtin_bit_cast(ws_f2_s, __builtin_amdgcn_raw_buffer_load_b64(wsr, (unsigned)lane * 8u, boff + u * 4u, 0));
    a = d.x; b = d.y;
  };
  auto st2f = [&](unsigned boff, unsigned u, float a, float b) __attribute__((always_inline)) {
    const ws_f2_s d = {a, b};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ws_v2u_s, d), wsr, (unsigned)lane * 8u, boff + u * 4u, 0);
  };
  // element (tix, r)[lane] of an Anderson array through its pointer (cold paths)
  auto apair = [&](unsigned tix, int r) -> size_t { return (size_t)tix * 256 + (size_t)((r >> 1) * 128 + lane * 2 + (r & 1)); };
  const unsigned kSiteTix = (unsigned)(NE * CT);   // tile index of site tile 0 in the Anderson arrays (NT / 256)

  // ---- init: inputs -> fragment order; |q|_inf, max ub; a session whose bounds cannot meet its energy row ----------
  real qn = 0, um = 0, bad = 0;
#pragma unroll 1
  for (int e = wave; e < NE; e += kStreamWaves) {
    RELANE();
    real lbv[CT][4], ubv[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
        const bool ok = ev < N && tt < Tm;
        const size_t idx = ((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0);
        const real l = ok ? A.lb[idx] : 0.0;
        real u = ok ? A.ub[idx] : 0.0;
        const real q = ok ? A.q[idx] : 0.0;
        if (u < l) u = l;
        lbv[c][r] = l; ubv[c][r] = u;
        LBs[fidx(e, c, r)] = l; UBs[fidx(e, c, r)] = u; Qs[fidx(e, c, r)] = q;
        qn = fmax(qn, fabs(q)); um = fmax(um, u);
      }
    if (flat_on) {   // (block-uniform) flat tiles: see tile_back
      auto bits = [](real v) -> long long { return __builtin_bit_cast(long long, v); };
      unsigned mk = 0; bool okl = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real L = lbv[0][r], U = ubv[0][r];
#pragma unroll
        for (int c = 1; c < CT; ++c) {   // the pair with the largest u (then l): the candidate "on" value of this lane's row
          const bool up = ubv[c][r] > U || (ubv[c][r] == U && lbv[c][r] > L);
          L = up ? lbv[c][r] : L; U = up ? ubv[c][r] : U;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {   // bit patterns: the rebuilt bounds are the stored ones to the last bit
          const bool on = bits(lbv[c][r]) == bits(L) && bits(ubv[c][r]) == bits(U), off = bits(lbv[c][r]) == 0 && bits(ubv[c][r]) == 0;
          okl = okl && (on || off);
          mk |= on && !off ? 1u << (r * CT + c) : 0u;
        }
        const unsigned o = cblu + (unsigned)(e * 4 + r) * 1024u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ws_v2u_s, L), wsr, (unsigned)lane * 16u, o, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ws_v2u_s, U), wsr, (unsigned)lane * 16u, o + 8u, 0);
      }
      __builtin_amdgcn_raw_buffer_store_b32(mk, wsr, (unsigned)lane * 4u, cbm + (unsigned)e * 256u, 0);
      const bool oktile = __builtin_amdgcn_ballot_w64(!okl) == 0;
      if (lane == 0) TileFlat[e] = oktile ? 1 : 0;   // read after the barriers of the block maximum below
    }
#pragma unroll 1
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ev = 16 * e + M::rowof(g, r);
        const size_t sidx = ((size_t)b * K + k) * N + (ev < N ? ev : 0);
        const int off = ev < N ? A.s_off[sidx] : 0, len = ev < N ? A.s_len[sidx] : 0;
        real sl = 0, su = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int tp = 16 * c + t;
          const bool inw = tp >= off && tp < off + len && tp < Tm;
          sl += inw ? lbv[c][r] : 0.0; su += inw ? ubv[c][r] : 0.0;
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
    // block-uniform doubles are forced into scalar registers (uniform_scalar, acn_qp_tiled.hpp): as vector-register
    // pairs a dozen of them were spilled around every tile of the fused pass
    qnorm = uniform_scalar(f[0]);
    pd = uniform_scalar(effective_pdiag<real>(pd_user, A.reg_rel, qnorm, f[1], A.horizon[b], lfb > 0.0 || dcb > 0.0));
    if (f[2] > 0) {
      for (size_t k = tid; k < (size_t)N * Tm; k += kStreamWaves * 64) A.x[(size_t)b * N * Tm + k] = 0;
      if (A.y_out)
        for (size_t k = tid; k < (size_t)A.Mg * Tm; k += kStreamWaves * 64) A.y_out[(size_t)b * A.Mg * Tm + k] = 0;
      if (tid == 0) { A.status[b] = 4; A.iters[b] = 0; A.pri[b] = M::big; A.dua[b] = M::big; A.obj[b] = 0; }
      break;   // (block-uniform) out of the pass loop: the next problem of the queue
    }
  }

  real rho = A.rho0;
  if (pass > 0) {   // fixed penalty retry_rho * 4^(pass - 1)
    rho = A.retry_rho;
    for (int k = 1; k < pass; ++k) rho *= 4.0;
    rho = uniform_scalar(rho);
  }
  // P = Ghat r0 (or Ghat z1 during the start): this wave's accumulators for the output tiles (m, c) it owns
  // wave m (< MT) owns the output tiles (m, c) of all column tiles c: a Ghat fragment is then fetched once per
  // iteration and row tile, not once per column tile as well (the fragments are the only re-read data of the kernel)
  static_assert(MT <= kStreamWaves, "one owner wave per row tile");
  constexpr int NOWN = CT;
  vec4 pown[NOWN];
  const int n_rounds = (NE + kStreamWaves - 1) / kStreamWaves;

  // ---- projection of ONE register row (one EVSE per 16-lane DPP row: EVSE 16 e + rowof(g, r), its periods = the 16
  // lanes times the CT column registers) onto B = box + energy rows.  Same safeguarded Newton as the general-shape
  // kernel / the C port.  One row at a time keeps only 3 * CT values live instead of 3 * 4 * CT. -----------------
  // `first`: the row's session slot 0 (window, capacity, last multiplier) fetched by the caller -- tile_back requests
  // the four rows' entries together with their bounds instead of one dependent L2 round trip per row
  struct Slot0 { int off, len; real cap, mu; bool have; };
  auto slot0_of = [&](int e, int r) __attribute__((always_inline)) -> Slot0 {
    const int ev = 16 * e + M::rowof(g, r);
    const size_t sidx = (size_t)b * K * N + (ev < N ? ev : 0);
    Slot0 s0;
    s0.off = ev < N ? A.s_off[sidx] : 0;
    s0.len = ev < N ? A.s_len[sidx] : 0;
    s0.cap = ev < N ? A.s_cap[sidx] : 0.0;
    s0.mu = ev < N ? MU[ev] : 0.0;
    s0.have = true;
    return s0;
  };
  auto project_row = [&](int e, int r, const real (&zh)[CT], const real (&lbv)[CT], const real (&ubv)[CT],
                         real (&z1)[CT], bool reset_mu, const Slot0 first) __attribute__((always_inline)) {
    const int ev = 16 * e + M::rowof(g, r);
#pragma unroll
    for (int c = 0; c < CT; ++c) z1[c] = fmin(fmax(zh[c], lbv[c]), ubv[c]);
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const size_t sidx = ((size_t)b * K + k) * N + (ev < N ? ev : 0);
      const bool pre = first.have && k == 0;   // uniform
      const int off = pre ? first.off : (ev < N ? A.s_off[sidx] : 0);
      int len = pre ? first.len : (ev < N ? A.s_len[sidx] : 0);
      if (off + len > Tm) len = Tm - off;
      const real cap = pre ? first.cap : (ev < N ? A.s_cap[sidx] : 0.0);
      real s0 = 0, sl = 0, su = 0, lo_l = M::big, hi_l = -M::big;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
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
      const real mu0 = (reset_mu || ev >= N) ? 0.0 : (pre ? first.mu : MU[(size_t)k * NP + ev]);
      real m = fmin(fmax(mu0, lo), hi);
#pragma unroll 1
      for (int guard = 0; guard <= 100; ++guard) {
        if (!__any(need)) break;
        real gl = 0, nl = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
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
      for (int c = 0; c < CT; ++c) {
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

  auto zero_pown = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NOWN; ++k) pown[k] = vec4{0, 0, 0, 0};
  };
  // One round of P: this wave's tile vector v (C layout = MFMA B operand; `have` = the wave has a tile this round)
  // goes into the slab; after the barrier the owner of each output tile (m, c) adds Ghat[m, e] v_e[c] for the tiles
  // e of the round, in tile order.  ONE barrier per round: the slabs alternate, and the slab of round rd is written
  // again in round rd + 2, after the barrier of round rd + 1, which an owner passes only with its reads of round rd
  // done.  p_finish() (one more barrier) closes a pass before anything else touches the slabs (h^ aliases the second).
  // (two halves: p_stage sits INSIDE the caller's `if (have)` next to the code that produced v -- with the slab write
  //  behind a second `if (have)` after the join the tile vector was live across the join as phi(v, undefined), and the
  //  allocator spilled and reloaded thirteen dead registers around every tile)
  auto p_stage = [&](int rd, const real (&v)[4][CT]) __attribute__((always_inline)) {
    real* RED = RED0 + (size_t)(rd & 1) * (kStreamWaves * CT * 256);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) RED[((wave * CT + c) * 4 + s) * 64 + lane] = v[s][c];
  };
  auto p_round = [&](int rd) __attribute__((always_inline)) {
    real* RED = RED0 + (size_t)(rd & 1) * (kStreamWaves * CT * 256);
    __syncthreads();
    const int ne = min(kStreamWaves, NE - rd * kStreamWaves);
    if (wave < MT) {
      const real* fg = FG + ((size_t)(rd * kStreamWaves) * MT + wave) * 2 * 4 * 64 + lane;
      // all fragments of the round are requested up front (<= 4 tiles x 4 k-steps), then the MFMA chains run
      real af[kStreamWaves][4];
#pragma unroll
      for (int w = 0; w < kStreamWaves; ++w)
#pragma unroll
        for (int s = 0; s < 4; ++s) af[w][s] = fg[(size_t)(w < ne ? w : 0) * MT * 2 * 4 * 64 + s * 64];
#pragma unroll
      for (int w = 0; w < kStreamWaves; ++w)
        if (w < ne) {
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) pown[c] = M::mma(af[w][s], RED[((w * CT + c) * 4 + s) * 64 + lane], pown[c]);
        }
    }
  };
  auto p_finish = [&]() __attribute__((always_inline)) { __syncthreads(); };
  // r0 of every tile from the stored state -> P (start, and after a rho change)
  auto rebuild_p = [&]() __attribute__((always_inline)) {
    zero_pown();
#pragma unroll 1
    for (int rd = 0; rd < n_rounds; ++rd) {
      RELANE();
      const int e = rd * kStreamWaves + wave;
      const bool have = e < NE;
      if (have) {
        real r0[4][CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t i = fidx(e, c, r);
            r0[r][c] = sigma * Xs[i] - Qs[i] + rho * Z1s[i] - Y1s[i];
          }
        p_stage(rd, r0);
      }
      p_round(rd);
    }
    p_finish();
  };

  // ---- start (see acn_qp_tiled.hpp): z1 = Proj_B(-kStartGain q), x = z1, y1 = -(q + pd z1); z2 = G z1 = Q (Ghat z1);
  // warm (optional): z1 = Proj_B(warm_x), y2 = warm_y, y1 = -(q + pd z1 + G' y2) -------------------------------------
  const bool warm = pass == 0 && A.warm_x != nullptr && A.warm_y != nullptr;
  zero_pown();
#pragma unroll 1
  for (int rd = 0; rd < n_rounds; ++rd) {
    RELANE();
    const int e = rd * kStreamWaves + wave;
    const bool have = e < NE;
    if (have) {
      real z1[4][CT];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real zs[CT], lbv[CT], ubv[CT], qv[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const size_t i = fidx(e, c, r);
          qv[c] = Qs[i]; lbv[c] = LBs[i]; ubv[c] = UBs[i];
          zs[c] = -kStartGain * qv[c];
          if (warm) {
            const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
            const bool ok = ev < N && tt < Tm;
            zs[c] = ok ? A.warm_x[((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0)] : 0.0;
          }
        }
        project_row(e, r, zs, lbv, ubv, z1[r], true, Slot0{0, 0, 0.0, 0.0, false});
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const size_t i = fidx(e, c, r);
          Xs[i] = z1[r][c]; Z1s[i] = z1[r][c]; Y1s[i] = -(qv[c] + pd * z1[r][c]);
        }
      }
      p_stage(rd, z1);
    }
    p_round(rd);
  }
  p_finish();
  // Ghat z1 sits with the owners: hand it to the site-row update below through G0H
  if (wave < MT) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) G0H[((wave * CT + c) * 4 + r) * 64 + lane] = pown[c][r];
  }
  __syncthreads();
#pragma unroll 1
  for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
      RELANE();
    const int mo = tl / CT, c = tl - mo * CT;
    vec4 zt = {0, 0, 0, 0};
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        zt = M::mma(FQ[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], G0H[((mi * CT + c) * 4 + s) * 64 + lane], zt);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ((mo * CT + c) * 4 + r) * 64 + lane;
      real yv = 0;
      if (warm) {
        const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
        const int ja = A.rowabi[j];
        if (ja >= 0 && tt < Tm) yv = A.warm_y[((size_t)b * A.Mg + ja) * Tm + tt] / static_cast<const real*>(A.rowscale)[j];
      }
      Z2[i] = zt[r]; GX[i] = zt[r]; Y2[i] = yv;
    }
  }
  __syncthreads();
  if (warm) {   // y1 = -(q + pd z1 + G' y2): the cold start stored the G' y2 = 0 version
#pragma unroll 1
    for (int e = wave; e < NE; e += kStreamWaves) {
      RELANE();
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        vec4 gty = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[((m * CT + c) * 4 + s) * 64 + lane], gty);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y1s[fidx(e, c, r)] -= gty[r];
      }
    }
  }
  if (warm) {
    rebuild_p();
  } else {
    // cold start: y1 = -(q + pd z1) and x = z1 make r0 = (sigma + pd + rho) z1 exactly, so P = a (Ghat z1) -- which the
    // owner waves still hold from the start rounds: one multiplication instead of a pass over four arrays
    const real a0 = sigma + pd + rho;
#pragma unroll
    for (int k = 0; k < NOWN; ++k) pown[k] = pown[k] * a0;
  }

  // u of the Anderson map from the current (z, y): after the start and after every rho change
  auto reset_u = [&]() __attribute__((always_inline)) {
    if (aa_m <= 0) return;
    ACNQP_STREAM_AA_PTRS();
    const real ir = 1.0 / rho;
#pragma unroll 1
    for (int e = wave; e < NE; e += kStreamWaves) {
      RELANE();
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const size_t i = fidx(e, c, r); UP[i] = Z1s[i] + Y1s[i] * ir; }   // same layout
    }
#pragma unroll 1
    for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
      RELANE();
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = (tl * 4 + r) * 64 + lane; UP[apair(kSiteTix + tl, r)] = Z2[i] + Y2[i] * ir; }
    }
  };
  reset_u();

  int status = 2, it = 0, n_adapt = 0, best_it = 0;
  real best_score = M::big;
  real pri = M::big, dua = M::big;
  bool done = false, have_prev = false;
  // Anderson state (block-uniform scalars, every wave keeps its own identical copy)
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  real* AaH = AaHS + (size_t)wave * (AMX * AMX + AMX);
  if (aa_m > 0)
    AaH[aa_clear_idx()] = 0;
  real sv0 = 0, sv2 = 0;   // |G x - z2|_inf, max(|G x|, |z2|): the site-row share of the residuals, per iteration

  // ---- projection of one site-row tile onto C from its pre-projection point: z2, y2, the tile's residual terms ------
  auto site_project = [&](int mo, int c, const real (&zhr)[4], const real* RLi, const int32_t* RTi) __attribute__((always_inline)) {
    const int tt = 16 * c + t;
    real pk = M::big;
    if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
    int ty[4];
    real lim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int j = 16 * mo + M::rowof(g, r); ty[r] = RTi[j]; lim[r] = RLi[j]; }
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
      const int i = ((mo * CT + c) * 4 + r) * 64 + lane;
      real zn = zhr[r];
      if (ty[r] == kRowBox) zn = fmin(zn, lim[r]);
      else if (ty[r] == kRowPeak) zn = fmin(zn, pk);
      else if (ty[r] == kRowQuad) zn = zn * (rho / (rho + lfb));
      else if (ty[r] == kRowSocRe || ty[r] == kRowSocIm) zn = zn * scl[r >> 1];
      // kRowMax: zn = zhr here; the horizon-wide prox (dc_row) follows once every column tile is through
      Y2[i] = rho * (zhr[r] - zn);
      Z2[i] = zn;
      if (!(dc_on && ty[r] == kRowMax)) {
        sv0 = fmax(sv0, fabs(GX[i] - zn));
        sv2 = fmax(sv2, fmax(fabs(GX[i]), fabs(zn)));
      }
    }
  };
  // ---- demand charge (acn_qp_tiled.hpp): z_t = min(zh_t, max(tau, floor)), tau = root of sum_t (zh_t - tau)+ = dc / rho,
  // over the whole horizon of the "max" row: ONE wave, after every site tile has left its pre-projection point of the
  // row in Z2.  The row's periods are the 16 lanes of one lane group (g = jdc & 3) x the column tiles.
  auto dc_row = [&]() __attribute__((always_inline)) {
    const int mo = jdc >> 4, rr = jdc & 15, gd = rr & 3, rd_ = rr >> 2;   // rowof(g, r) = g + 4 r
    real zv[CT], gxv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) { const int i = ((mo * CT + c) * 4 + rd_) * 64 + lane; zv[c] = Z2[i]; gxv[c] = GX[i]; }
    const real cw = dcb / rho;
    real vmax_l = -M::big;
#pragma unroll
    for (int c = 0; c < CT; ++c) vmax_l = (16 * c + t < Tm) ? fmax(vmax_l, zv[c]) : vmax_l;
    const real vmax = row_max<real>(vmax_l);
    real tau = vmax - cw;
    bool need = g == gd;
    int guard = 0;
    while (__any(need)) {
      ++guard;
      real sl = 0, nl = 0;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const real dd = zv[c] - tau;
        const bool on = (16 * c + t < Tm) && dd > 0.0;
        sl += on ? dd : 0.0;
        nl += on ? 1.0 : 0.0;
      }
      const real S = row_sum<real>(sl), nn = row_sum<real>(nl);
      const real f = S - cw;
      const real tn = nn > 0.0 ? tau + f / nn : vmax - cw;
      const bool fin = fabs(f) <= M::proj_tol * fmax(1.0, cw) * 16.0 || tn == tau || guard > 200;
      tau = (need && !fin) ? tn : tau;
      need = need && !fin;
    }
    const real lev = fmax(tau, dfl);
    if (g == gd) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int i = ((mo * CT + c) * 4 + rd_) * 64 + lane;
        const real zn = (16 * c + t < Tm) ? fmin(zv[c], lev) : zv[c];
        Y2[i] = rho * (zv[c] - zn);
        Z2[i] = zn;
        sv0 = fmax(sv0, fabs(gxv[c] - zn));
        sv2 = fmax(sv2, fmax(fabs(gxv[c]), fabs(zn)));
      }
    }
  };
  // ---- the two halves of the fused tile pass.  front: load (x, z1, y1, q), r0, x~ (MFMA, e^ from LDS), relaxation ->
  // the pre-projection point zh and sq = sigma x_new - q (all the new r0 still needs of x and q); x_new is stored.
  auto tile_front = [&](int e, real inv_a, real inv_rho, real (&zh)[4][CT], real (&sq)[4][CT]) __attribute__((always_inline)) {
    const real* fg = FG + (size_t)e * MT * 2 * 4 * 64;
    real fx[MT][4];              // A fragments of the x~ product: requested with the state, used after it arrived
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s = 0; s < 4; ++s) fx[m][s] = fg[((m * 2 + 1) * 4 + s) * 64 + lane];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      real xv[4], z1o[4], y1o[4], qv[4];
      vec4 acc;
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const unsigned i = fidp(e, c, rp);
        ld2_nt(oX, i, xv[2 * rp], xv[2 * rp + 1]); ld2_nt(oZ1, i, z1o[2 * rp], z1o[2 * rp + 1]);
        ld2_nt(oY1, i, y1o[2 * rp], y1o[2 * rp + 1]); ld2_nt(oQ, i, qv[2 * rp], qv[2 * rp + 1]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sigma * xv[r] - qv[r] + rho * z1o[r] - y1o[r];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc = M::mma(fx[m][s], WE[((m * CT + c) * 4 + s) * 64 + lane], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real xn = acc[r] * inv_a;
        zh[r][c] = alpha * xn + (1.0 - alpha) * z1o[r] + y1o[r] * inv_rho;
        xv[r] = alpha * xn + (1.0 - alpha) * xv[r];   // x_new
        sq[r][c] = sigma * xv[r] - qv[r];
        asm volatile("" : "+v"(sq[r][c]));   // formed HERE: the compiler otherwise keeps x_new and q (spilled) until the new r0
      }
      st2_nt(oX, fidp(e, c, 0), xv[0], xv[1]);
      st2_nt(oX, fidp(e, c, 1), xv[2], xv[3]);
#if ACNQP_STREAM_FRONT_FENCE
      __builtin_amdgcn_sched_barrier(0);   // one column tile's 16 loads in flight at a time (see the note at the top)
#endif
    }
  };
  // back: box + energy-row projection of zh one register row at a time, z1 and y1 stored, the tile's NEW r0 left in zh
  auto tile_back = [&](int e, real (&zh)[4][CT], const real (&sq)[4][CT]) __attribute__((always_inline)) {
    // the session slots of all four register rows are requested at once; the bounds one register PAIR at a time, right
    // before that pair's rows are projected: 24 registers held instead of 48, which takes 7 of the 11 spill stores and 6
    // of the 20 reloads out of the tile loop (272 -> 208 B of scratch per lane; configs[4] leg 421.8 -> 417.4 ms, A/B in
    // one session) at the price of a second dependent round trip per tile.  (Requesting all of them BEFORE tile_front,
    // under its loads and MFMA chain, was measured too: 468 -> 507 ms -- the 64 registers they pin spill the front.)
    // Flat tiles.  The rate bounds of a session are two constants inside its window and zero outside (aco.py:45-73), so
    // per lane the CT (l, u) pairs of a row nearly always take ONE non-zero value: the init phase checks exactly that and
    // keeps (l, u) per row + one word of period bits per lane; such a tile's bounds are rebuilt from 68 bytes per lane
    // instead of streamed (2 x 4 x CT x 8) -- bit for bit the same numbers.  Any other tile streams them as before.
    const bool flat = flat_on && __builtin_amdgcn_readfirstlane((int)TileFlat[e]) != 0;
    ws_d2_s lu[4];
    unsigned mk = 0;
    if (flat) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lu[r] = __builtin_bit_cast(ws_d2_s, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, cblu + (unsigned)(e * 4 + r) * 1024u, 0));
      mk = __builtin_amdgcn_raw_buffer_load_b32(wsr, (unsigned)lane * 4u, cbm + (unsigned)e * 256u, 0);
    }
    Slot0 s4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s4[r] = slot0_of(e, r);
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      real lbp[2][CT], ubp[2][CT];
      if (flat) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const bool on = (mk >> ((2 * rp + h) * CT + c)) & 1u;
            lbp[h][c] = on ? lu[2 * rp + h].x : 0.0; ubp[h][c] = on ? lu[2 * rp + h].y : 0.0;
          }
      } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const unsigned i = fidp(e, c, rp);
          ld2_nt(oLB, i, lbp[0][c], lbp[1][c]); ld2_nt(oUB, i, ubp[0][c], ubp[1][c]);
        }
      }
      real z1a[CT], z1b[CT];
      project_row(e, 2 * rp, zh[2 * rp], lbp[0], ubp[0], z1a, false, s4[2 * rp]);
      project_row(e, 2 * rp + 1, zh[2 * rp + 1], lbp[1], ubp[1], z1b, false, s4[2 * rp + 1]);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const unsigned i = fidp(e, c, rp);
        const real ya = rho * (zh[2 * rp][c] - z1a[c]), yb = rho * (zh[2 * rp + 1][c] - z1b[c]);
        st2_nt(oZ1, i, z1a[c], z1b[c]); st2_nt(oY1, i, ya, yb);
        zh[2 * rp][c] = sq[2 * rp][c] + rho * z1a[c] - ya;
        zh[2 * rp + 1][c] = sq[2 * rp + 1][c] + rho * z1b[c] - yb;
      }
    }
  };
  // Residual terms of the iteration's check: per wave in LDS (ResS), not in registers -- as five loop-carried maxima
  // (plus the site rows' two) they lived across the tile loop of EVERY iteration for the sake of one iteration in twenty,
  // and the allocator spilled and reloaded them around every tile (4 of the tile loop's spill stores and 8 of its
  // reloads).  A maximum does not care in which order or grouping it is taken: same bits.
  auto fold_site_residuals = [&](bool chk) __attribute__((always_inline)) {   // before a tile loop: sv0 / sv2 end here
    if (chk) {
      const real m0 = wave_max<real>(sv0), m2 = wave_max<real>(sv2);
      if (lane == 0) { ResS[wave * 8 + 0] = m0; ResS[wave * 8 + 1] = 0; ResS[wave * 8 + 2] = m2; ResS[wave * 8 + 3] = 0; ResS[wave * 8 + 4] = 0; }
    }
  };
  auto tile_residuals = [&](int e) __attribute__((always_inline)) {
    // residual terms of this tile (state re-read: L2-hot); (G' y2) tile by MFMA with the un-rotated site matrix
    real v0 = 0, v1 = 0, v2 = 0, v4 = 0, v5 = 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      vec4 gty = {0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[((m * CT + c) * 4 + s) * 64 + lane], gty);
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
    v0 = wave_max<real>(v0); v1 = wave_max<real>(v1); v2 = wave_max<real>(v2); v4 = wave_max<real>(v4); v5 = wave_max<real>(v5);
    if (lane == 0) {
      real* R = ResS + wave * 8;
      R[0] = fmax(R[0], v0); R[1] = fmax(R[1], v1); R[2] = fmax(R[2], v2); R[3] = fmax(R[3], v4); R[4] = fmax(R[4], v5);
    }
  };
#ifdef ACNQP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
#pragma unroll 1
  while (!done) {
    ++it;
    const real a = sigma + pd + rho, inv_a = uniform_scalar(1.0 / a), inv_rho = uniform_scalar(1.0 / rho);
    const bool check = (it % A.check_every == 0) || it >= max_iter_p;
    const bool ev_it = aa_m > 0 && it % kAaPeriod == 0;   // Anderson event: the projections follow the extrapolation
    // an offset the compiler cannot see through keeps the loads of loop-invariant site data (Q fragments, row
    // constants) inside the loop, where they hit L1 / L2, instead of pinning ~150 registers across it
    unsigned zoff = 0;
    asm volatile("" : "+s"(zoff));
    const real* FQi = FQ + zoff;
    const real* Lmi = Lm + zoff;
    const real* RLi = RL + zoff;
    const int32_t* RTi = A.rowtype + zoff;
    sv0 = 0; sv2 = 0;
    // ---- eigen space, by the wave that owns each 16 x 16 site tile (its accumulator holds Ghat r0 for that tile):
    // e^ -> WE, h^ -> G0H ---------------------------------------------------------------------------------------
    // (every operand of a product is requested before its MFMA chain starts: left alone the compiler emits load,
    //  s_waitcnt vmcnt(0), MFMA thirty-six times in a row, each an L2 round trip under the load of the streaming waves)
    if (wave < MT) {
      RELANE();
      const int mo = wave;
      real fq[MT][4];   // Q' fragments of this row tile: the same for every column tile
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) fq[mi][s] = FQi[(((mo * MT + mi) * 2 + 0) * 4 + s) * 64 + lane];
#pragma unroll
      for (int k = 0; k < NOWN; ++k) {
        const int c = k;
        real z2v[MT][4], y2v[MT][4], ljv[4];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int i = ((mi * CT + c) * 4 + s) * 64 + lane;
            z2v[mi][s] = Z2[i]; y2v[mi][s] = Y2[i];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) ljv[r] = Lmi[16 * mo + M::rowof(g, r)];
        __builtin_amdgcn_sched_barrier(0);
        vec4 wh = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) wh = M::mma(fq[mi][s], rho * z2v[mi][s] - y2v[mi][s], wh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ((mo * CT + c) * 4 + r) * 64 + lane;
          const real lj = ljv[r];
          const real g0 = pown[k][r];
          const real e_ = wh[r] - (rho / (a + rho * lj)) * (g0 + lj * wh[r]);
          WE[i] = e_;
          G0H[i] = (g0 + lj * e_) * inv_a;
        }
      }
    }
    __syncthreads();
    // ---- site rows: G x~ = Q h^, relaxation; projection onto C and y2 (owner waves) -- after the event on event iterations
#pragma unroll 1
    for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
      RELANE();
      const int mo = tl / CT, c = tl - mo * CT;
      real fq[MT][4], hv[MT][4], gxo[4], z2o[4], y2o[4];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          fq[mi][s] = FQi[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane];
          hv[mi][s] = G0H[((mi * CT + c) * 4 + s) * 64 + lane];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ((mo * CT + c) * 4 + r) * 64 + lane;
        gxo[r] = GX[i]; z2o[r] = Z2[i]; y2o[r] = Y2[i];
      }
      __builtin_amdgcn_sched_barrier(0);
      vec4 zt = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) zt = M::mma(fq[mi][s], hv[mi][s], zt);
      real zhr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ((mo * CT + c) * 4 + r) * 64 + lane;
        GX[i] = alpha * zt[r] + (1.0 - alpha) * gxo[r];
        zhr[r] = alpha * zt[r] + (1.0 - alpha) * z2o[r] + y2o[r] * inv_rho;
      }
      if (ev_it) {
        st2(oZH, (kSiteTix + tl) * 256u, zhr[0], zhr[1]);
        st2(oZH, (kSiteTix + tl) * 256u + 128u, zhr[2], zhr[3]);
      } else {
        site_project(mo, c, zhr, RLi, RTi);
      }
    }
    __syncthreads();
    STAMP(0);   // eigen step + site rows + their two barriers
    if (!ev_it) {
      if (dc_on && wave == kStreamWaves - 1) { RELANE(); dc_row(); }
      fold_site_residuals(check);
      // ---- the fused pass over this wave's EVSE tiles -----------------------------------------------------------
      zero_pown();
#pragma unroll 1
      for (int rd = 0; rd < n_rounds; ++rd) {
        RELANE();
        const int e = rd * kStreamWaves + wave;
        const bool have = e < NE;
        if (have) {
          real zh[4][CT], sq[4][CT];
          tile_front(e, inv_a, inv_rho, zh, sq);
          STAMP(1);   // fused pass: front (loads, MFMA, x store)
          tile_back(e, zh, sq);
          STAMP(2);   // fused pass: back (bounds, water-filling, z1 / y1 stores)
          if (check) tile_residuals(e);
          STAMP(3);   // residual terms (check iterations)
          p_stage(rd, zh);   // the tile's new r0 joins next iteration's P
        }
        p_round(rd);
        STAMP(4);   // round: slab, barrier, owners' MFMA
      }
      p_finish();
      STAMP(5);   // closing barrier
    } else {
      // ================= Anderson event (type II; acn_qp_tiled.hpp / oracle/admm_port.c) ============================
      // u = (zh, zhr) is the state of the fixed-point map.  Pass A: every tile's zh -> ZHs, the new column pair
      // (dF, dG) -> ring slot, the dot products dF_new . dF_j, dF_new . f, f . f.
      const bool col = aa_have_prev;
      const int slot = aa_head;
      real d[AMX + 2];
#pragma unroll
      for (int j = 0; j < AMX + 2; ++j) d[j] = 0;
      // the four registers of one tile column: g = gv[r], state index uo + 64 r
      auto aa_tile = [&](const real (&gv)[4], unsigned tix) __attribute__((always_inline)) {
        real uv[4], fpv[4], gpv[4];
        float hv[AMX][4];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const unsigned u = tix * 256u + (unsigned)rp * 128u;
          ld2(oUP, u, uv[2 * rp], uv[2 * rp + 1]); ld2(oFP, u, fpv[2 * rp], fpv[2 * rp + 1]); ld2(oGP, u, gpv[2 * rp], gpv[2 * rp + 1]);
        }
        // every ring column is requested, live or not (a branch per column would put each load in a basic block of its
        // own: five dependent memory round trips per tile instead of one); a dead column's data is replaced by zeros
#pragma unroll
        for (int j = 0; j < AMX; ++j) {
          const unsigned jj = (unsigned)(j < aa_m ? j : aa_m - 1);
          ld2f(bHF + jj * (unsigned)DU * 4u, tix * 256u, hv[j][0], hv[j][1]);
          ld2f(bHF + jj * (unsigned)DU * 4u, tix * 256u + 128u, hv[j][2], hv[j][3]);
        }
#pragma unroll
        for (int j = 0; j < AMX; ++j) {
          const bool live = ((aa_valid >> j) & 1u) && j != slot;   // uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[j][r] = live ? hv[j][r] : 0.f;
        }
        float cqv[4], cgv[4];
        real fv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real f = gv[r] - uv[r];
          d[AMX + 1] += f * f;
          const float cq = (float)(f - fpv[r]);
          const float cg = (float)(gv[r] - gpv[r]);
#pragma unroll
          for (int j = 0; j < AMX; ++j) d[j] += (real)cq * (j == slot ? (real)cq : (real)hv[j][r]);
          d[AMX] += (real)cq * f;
          cqv[r] = cq; cgv[r] = cg; fv[r] = f;
        }
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const unsigned u = tix * 256u + (unsigned)rp * 128u;
          if (col) {
            st2f(bHF + (unsigned)slot * (unsigned)DU * 4u, u, cqv[2 * rp], cqv[2 * rp + 1]);
            st2f(bHG + (unsigned)slot * (unsigned)DU * 4u, u, cgv[2 * rp], cgv[2 * rp + 1]);
          }
          st2(oFP, u, fv[2 * rp], fv[2 * rp + 1]); st2(oGP, u, gv[2 * rp], gv[2 * rp + 1]);
        }
      };
#pragma unroll 1
      for (int rd = 0; rd < n_rounds; ++rd) {
        RELANE();
        const int e = rd * kStreamWaves + wave;
        if (e < NE) {
          real zh[4][CT], sq[4][CT];
          tile_front(e, inv_a, inv_rho, zh, sq);
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            real gv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = zh[r][c];
            st2(oZH, (unsigned)(e * CT + c) * 256u, gv[0], gv[1]);
            st2(oZH, (unsigned)(e * CT + c) * 256u + 128u, gv[2], gv[3]);
            aa_tile(gv, (unsigned)(e * CT + c));
          }
        }
      }
#pragma unroll 1
      for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
        RELANE();
        real gv[4];
        ld2(oZH, (kSiteTix + tl) * 256u, gv[0], gv[1]);
        ld2(oZH, (kSiteTix + tl) * 256u + 128u, gv[2], gv[3]);
        aa_tile(gv, kSiteTix + (unsigned)tl);
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
        for (int wv = 0; wv < kStreamWaves; ++wv) sw += AaRedS[wv * (AMX + 2) + j];
        d[j] = sw;
      }
      const real fn = sqrt(d[AMX + 1]);
      bool keep = col;
      if (aa_was && fn > kAaSafe * fn_prev) {
        // the accelerated step made the residual worse: clear the ring, back off exponentially
        aa_cnt = 0; aa_head = 0; aa_valid = 0; keep = false;
        __builtin_amdgcn_wave_barrier();
        AaH[aa_clear_idx()] = 0;
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
          const real piv = lane_value(ae, 9 * k);
          const real rk = __shfl(ae, 8 * k + gj);
          const real ck = __shfl(ae, 8 * gi + k);
          const real rs = rk / piv;
          ae = gi == k ? rs : ae - ck * rs;
        }
#pragma unroll
        for (int j = 0; j < AMX; ++j) gam[j] = lane_value(ae, 8 * j + AMX);
        aa_was = true;
      }
      // u = g - sum_j gamma_j dG_j for the four registers at uo; stored as the new u
      auto aa_apply = [&](real (&out)[4], unsigned tix) __attribute__((always_inline)) {
        if (ext) {
          float hg[AMX][4];
#pragma unroll
          for (int j = 0; j < AMX; ++j) {   // all columns in one batch (see aa_tile); dead ones are skipped below
            const unsigned jj = (unsigned)(j < aa_m ? j : aa_m - 1);
            ld2f(bHG + jj * (unsigned)DU * 4u, tix * 256u, hg[j][0], hg[j][1]);
            ld2f(bHG + jj * (unsigned)DU * 4u, tix * 256u + 128u, hg[j][2], hg[j][3]);
          }
#pragma unroll
          for (int j = 0; j < AMX; ++j) {
            const bool live = (aa_valid >> j) & 1u;   // uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = live ? out[r] - gam[j] * (real)hg[j][r] : out[r];
          }
        }
        st2(oUP, tix * 256u, out[0], out[1]);
        st2(oUP, tix * 256u + 128u, out[2], out[3]);
      };
      // ---- the site rows are projected from their (extrapolated) point ------------------------------------------
#pragma unroll 1
      for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
        RELANE();
        const int mo = tl / CT, c = tl - mo * CT;
        real zhr[4];
        ld2(oZH, (kSiteTix + tl) * 256u, zhr[0], zhr[1]);
        ld2(oZH, (kSiteTix + tl) * 256u + 128u, zhr[2], zhr[3]);
        aa_apply(zhr, kSiteTix + (unsigned)tl);
        site_project(mo, c, zhr, RLi, RTi);
      }
      __syncthreads();
      if (dc_on && wave == kStreamWaves - 1) { RELANE(); dc_row(); }
      fold_site_residuals(check);
      // ---- pass B: the extrapolated zh of every tile -> projection, y1, the new r0 -> P ----------------------------
      zero_pown();
#pragma unroll 1
      for (int rd = 0; rd < n_rounds; ++rd) {
        RELANE();
        const int e = rd * kStreamWaves + wave;
        const bool have = e < NE;
        if (have) {
          real zh[4][CT], sq[4][CT];
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            real o4[4], xq[4], qq[4];
            const unsigned tix = (unsigned)(e * CT + c);
            ld2(oZH, tix * 256u, o4[0], o4[1]); ld2(oZH, tix * 256u + 128u, o4[2], o4[3]);
            ld2_nt(oX, fidp(e, c, 0), xq[0], xq[1]); ld2_nt(oX, fidp(e, c, 1), xq[2], xq[3]);   // x_new was stored by pass A
            ld2_nt(oQ, fidp(e, c, 0), qq[0], qq[1]); ld2_nt(oQ, fidp(e, c, 1), qq[2], qq[3]);
            aa_apply(o4, tix);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              zh[r][c] = o4[r];
              sq[r][c] = sigma * xq[r] - qq[r];
            }
          }
          tile_back(e, zh, sq);
          if (check) tile_residuals(e);
          p_stage(rd, zh);
        }
        p_round(rd);
      }
      p_finish();
      STAMP(6);   // Anderson event iteration (both passes)
    }
    if (check) {
      unsigned soff_ = 0;
      asm volatile("" : "+s"(soff_));
      float* Y1P = reinterpret_cast<float*>(W0 + off_snap + soff_);
      float* Y2P = Y1P + NT;
      real v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = ResS[wave * 8 + k];   // (site rows folded in before the tile loop)
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
        for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
          RELANE();
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int i = (tl * 4 + r) * 64 + lane; w6[0] = fmax(w6[0], fabs(Y2[i] - (real)Y2P[i])); }
        }
#pragma unroll 1
        for (int e = wave; e < NE; e += kStreamWaves) {
          RELANE();
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            vec4 gtv = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                const int i = ((m * CT + c) * 4 + s) * 64 + lane;
                gtv = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], Y2[i] - (real)Y2P[i], gtv);
              }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const size_t i = fidx(e, c, r);
              const real v1_ = Y1s[i] - (real)Y1P[i];
              w6[0] = fmax(w6[0], fabs(v1_));
              w6[1] = fmax(w6[1], fabs(v1_ + gtv[r]));
            }
          }
        }
        stream_block_max<2, NWV>(w6, SC, lane, wave);
        const real vn = w6[0];
        const real vtol = 1e-4 * vn;
        if (vn > 1e-12 * fmax(1.0, qnorm) && w6[1] <= vtol) {   // block-uniform
          real ssum = 0, badv = 0;
#pragma unroll 1
          for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {   // site rows
            RELANE();
            const int m = tl / CT, c = tl - m * CT;
            const int tt = 16 * c + t;
            real pk = M::big;
            if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = (tl * 4 + r) * 64 + lane;
              const int j = 16 * m + M::rowof(g, r);
              const real v2_ = Y2[i] - (real)Y2P[i];
              const int ty = RTi[j];
              if (ty == kRowBox) { ssum += RLi[j] * fmax(v2_, 0.0); if (v2_ < -vtol) badv = 1; }
              else if (ty == kRowPeak) {
                if (pk < M::big) ssum += pk * fmax(v2_, 0.0); else if (v2_ > vtol) badv = 1;
                if (v2_ < -vtol) badv = 1;
              } else if (ty == kRowSocRe) {
                const int i2 = (tl * 4 + ((r + 1) & 3)) * 64 + lane;
                const real vi = Y2[i2] - (real)Y2P[i2];
                ssum += RLi[j] * sqrt(v2_ * v2_ + vi * vi);
              } else if (ty == kRowSocIm) {
              } else if (fabs(v2_) > vtol) badv = 1;   // free / prox rows admit no ray
            }
          }
          // sessions (one register row at a time): bound each session's support function; periods outside every
          // window are pinned to lb = ub: support lb * v
#pragma unroll 1
          for (int ri = wave; ri < 4 * NE; ri += kStreamWaves) {
            RELANE();
            const int e = ri >> 2, r = ri & 3;
            const int ev = 16 * e + M::rowof(g, r);
            real vv[CT], lbv[CT], ubv[CT];
            bool cov[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              const size_t i = fidx(e, c, r);
              vv[c] = Y1s[i] - (real)Y1P[i]; lbv[c] = LBs[i]; ubv[c] = UBs[i]; cov[c] = false;
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
              for (int c = 0; c < CT; ++c) {
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
                for (int c = 0; c < CT; ++c) {
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
            for (int c = 0; c < CT; ++c)
              if (!cov[c] && 16 * c + t < Tm && ev < N) ssum += lbv[c] * vv[c];
          }
          const real tot = wave_sum<real>(ssum);
          real bd[1] = {badv};
          stream_block_max<1, NWV>(bd, SC, lane, wave);
          if (lane == 0) RED0[wave] = tot;   // the rounds' slab is free between iterations
          __syncthreads();
          real stot = 0;
          for (int wv = 0; wv < kStreamWaves; ++wv) stot += RED0[wv];
          __syncthreads();
          if (bd[0] == 0.0 && stot < -vtol) { status = 3; done = true; }
        }
      }
      if (!done) {   // snapshot for the next certificate test (single precision: acn_qp_tiled.hpp)
#pragma unroll 1
        for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
          RELANE();
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int i = (tl * 4 + r) * 64 + lane; Y2P[i] = (float)Y2[i]; }
        }
#pragma unroll 1
        for (int e = wave; e < NE; e += kStreamWaves) {
          RELANE();
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const size_t i = fidx(e, c, r); Y1P[i] = (float)Y1s[i]; }
        }
        have_prev = true;
      }
      const real score = fmax(pri / fmax(eps_p, 1e-300), dua / fmax(eps_d, 1e-300));
      if (score < kStallGain * best_score) { best_score = score; best_it = it; }
      const bool inacc = inaccurate_ok<real>(pri, dua, npri, ndua, A.eps_abs, A.eps_rel, A.inacc_floor);
      const bool stalled = A.stall_iters > 0 && it - best_it >= A.stall_iters && score <= kStallNear * best_score;   // acn_qp_tiled.hpp
      if (done) {
      } else if (it >= max_iter_p || stalled) {
        done = true;
        if (inacc) status = 5;
      } else if (adapt_p > 0 && it % adapt_p == 0) {
        const real sp = pri / fmax(npri, 1e-12), sd = dua / fmax(ndua, 1e-12);
        const real ratio = sqrt(sp / fmax(sd, 1e-30));
        const real tol_eff = A.adapt_tol * (1.0 + (real)n_adapt * (1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = uniform_scalar(fmin(fmax(rho * ratio, 1e-6), 1e6));
          __syncthreads();     // every wave's stores of this pass are visible before the state is re-read
          rebuild_p();         // r0 depends on rho: P with the new penalty
          if (aa_m > 0) {      // the fixed-point map changed: restart the ring from the current (z, y)
            reset_u();
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
            __builtin_amdgcn_wave_barrier();
            AaH[aa_clear_idx()] = 0;
          }
        }
      }
    }
  }

  // ---- results of this pass: the feasible iterate z1 is the schedule (kept if it beats the earlier passes) ------
  it_total += it;
  __syncthreads();
  if (pass == 0 || status_rank(status) > status_rank(best_status)) {   // block-uniform
  best_status = status;
  real ol = 0;
#pragma unroll 1
  for (int e = wave; e < NE; e += kStreamWaves)
#pragma unroll
    for (int c = 0; c < CT; ++c)
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
  if (A.y_out) {   // site-row multipliers in the caller's row order and units, by the waves that own the tiles
#pragma unroll 1
    for (int tl = wave; tl < MT * CT; tl += kStreamWaves) {
      const int mo = tl / CT, c = tl - mo * CT;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
        const int ja = A.rowabi[j];
        if (ja >= 0 && tt < Tm)
          A.y_out[((size_t)b * A.Mg + ja) * Tm + tt] = Y2[((mo * CT + c) * 4 + r) * 64 + lane] * static_cast<const real*>(A.rowscale)[j];
      }
    }
  }
  ol = wave_sum<real>(ol);
  if (lane == 0) SC[wave] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < kStreamWaves; ++wv) o += SC[wv];
    A.status[b] = status; A.pri[b] = pri; A.dua[b] = dua; A.obj[b] = o;
  }
  }
  if (tid == 0) A.iters[b] = it_total;
#ifdef ACNQP_STAMPS
  STAMP(7);   // residual check, certificate, penalty update, output
  if (lane == 0 && b < 1024 && wave < 16)
    for (int k = 0; k < 12; ++k) g_stamps[(b * 16 + wave) * 12 + k] = st_acc[k];
#endif
  if (!retry_wanted(pass, A.retry_passes, status, it, A.stall_iters, A.adapt_every)) break;
  __syncthreads();
  }   // passes
  }   // work queue
}

#undef RELANE
#undef ACNQP_STREAM_AA_PTRS
}  // namespace acnqp
