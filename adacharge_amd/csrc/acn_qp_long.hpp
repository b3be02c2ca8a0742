// Long-horizon kernel of the batched MPC QP solver for gfx950: horizons of 33 ... 288 periods (the reference's
// N = 54 x T = 144 stress scenarios, tests/test_adacharge_stress.py; day-long offline problems, aco.py:403-408),
// any site of up to 32 padded rows -- the shapes neither the register-resident kernel (T <= 32) nor the large-site
// kernel (T <= 48, whole rows of every tile in registers AND LDS slabs sized by the horizon) can hold.
//
// Same ADMM as acn_qp_tiled.hpp / acn_qp_stream.hpp.  What changes is who owns what.  Only ONE step of the iteration
// couples the periods of an EVSE: the projection onto its energy rows (water-filling over the session window).
// Everything else -- P = Ghat r0, the eigen-space step, the site-row projection, x~ = (r0 + Ghat' e^) / a -- is
// independent per period.  An iteration is three phases over small work items, dealt round-robin to the 8 waves
// of the problem's workgroup, with the state streaming through a per-problem workspace in MFMA fragment order
// (L2 / MALL resident at these sizes: 54 x 144 is 74 KB per array):
//
//   1a  item = one 16 x 16 SITE tile (row tile m, column tile c):
//         P = sum_e Ghat[m, e] r0[e, c],  w^ = Q[:, m]' (rho z2 - y2)[:, c]   (MFMA)  ->  e^, h^ of the tile -> LDS
//   1b  item = one site tile:  G x~ = Q h^ (MFMA, h^ from LDS), relaxation, projection onto C, y2
//       item = one EVSE tile (e, c):  x~ = (r0 + Ghat[:, e]' e^) / a (MFMA, e^ from LDS), relaxation, zh -> workspace
//   2   item = one REGISTER ROW of an EVSE tile (4 EVSEs x the WHOLE horizon, CTL column registers per lane):
//         zh, lb, ub -> water-filling (safeguarded Newton along the 16-lane DPP rows) -> z1, y1, the new r0
//
// Three barriers per iteration whatever the horizon; e^ and h^ (the only data every tile of a column needs) stay in
// LDS.  A 54 x 144 problem has 9-18 items in 1a, 45-54 in 1b, 16 in phase 2: every phase keeps all four SIMDs busy.
// What bounds an iteration is the CU's vector-memory issue rate: a 64-lane 8-byte access costs ~16 cycles of the
// texture addresser whoever issues it (tools/micro/rt_latency.hip), and an iteration makes ~2,800 of them.
// Anderson acceleration as in the other kernels (the ring lives in the workspace; its passes run over the tile items
// of 1b); the infeasibility certificate of acn_qp_tiled.hpp; the demand-charge row (its prox couples the periods of a
// SITE row: one more item, run by one wave at the start of phase 2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "acn_qp_stream.hpp"

namespace acnqp {

// The per-problem workspace is addressed as ONE buffer resource: a 128-bit descriptor in scalar registers, the lane's
// byte offset in one vector register shared by every access, and the array / tile offset as a 32-bit scalar operand
// of the instruction.  With ~20 arrays, plain pointers cost a 64-bit per-lane address per array (the compiler forms
// base + lane once and keeps -- or spills -- all of them); this costs nothing per access.
typedef unsigned ws_v2u __attribute__((ext_vector_type(2)));
typedef unsigned ws_v4u __attribute__((ext_vector_type(4)));
typedef double ws_d2 __attribute__((ext_vector_type(2)));
typedef float ws_f2 __attribute__((ext_vector_type(2)));
struct WsArr64 { unsigned off; };   // byte offset of an array of doubles inside the problem's workspace
struct WsArr32 { unsigned off; };   // ... of floats
__device__ inline WsArr32 operator+(WsArr32 a, size_t n) { return WsArr32{a.off + (unsigned)n * 4u}; }
struct WsRef64 {
  __amdgpu_buffer_rsrc_t rs; unsigned vo, so;
  __device__ inline operator double() const { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0)); }
  __device__ inline void operator=(double v) const { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ws_v2u, v), rs, vo, so, 0); }
  __device__ inline void operator-=(double v) const { *this = (double)*this - v; }
};
struct WsRef32 {
  __amdgpu_buffer_rsrc_t rs; unsigned vo, so;
  __device__ inline operator float() const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0)); }
  __device__ inline void operator=(float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, so, 0); }
};

// lane-offset flag of a workspace access that must not happen: beyond num_records of any problem's descriptor (a few
// MB), so the buffer unit answers a load with zeros and drops a store without touching memory
#ifndef ACNQP_LONG_PAD_SKIP
#define ACNQP_LONG_PAD_SKIP 1
#endif
constexpr unsigned kWsKill = ACNQP_LONG_PAD_SKIP ? 0x80000000u : 0u;
#ifndef ACNQP_LONG_FLAT_BOUNDS
#define ACNQP_LONG_FLAT_BOUNDS 1
#endif
constexpr int kLongAccelMax = 5;
#ifndef ACNQP_LONG_PAIR_ROWS
#define ACNQP_LONG_PAIR_ROWS 9
#endif
constexpr int kLongPairRows = ACNQP_LONG_PAIR_ROWS;   // longest row (column tiles) whose two paired rows fit the registers   // Anderson columns (the ring lives in the workspace: any shape takes all five)

// Compressed bounds of a row item (two register rows x 64 lanes): per row and lane one (l, u) pair of doubles, and per
// lane the two rows' period masks -- 40 bytes per lane instead of 2 x 2 x CTL x 8 (see `flat items` in the kernel)
constexpr int kLongFlatRows = 256;   // rows (of 4 EVSEs) whose flags fit the LDS table: sites of up to 1,024 EVSEs
__host__ __device__ inline long long long_flat_doubles(int NP) { return (long long)(NP / 16) * 640; }

// doubles of workspace one problem needs (accel = Anderson columns in use)
__host__ __device__ inline long long long_workspace(int NP, int CTL, int K, int MT, int accel) {
  const long long NT = (long long)(NP / 16) * CTL * 256, MS = (long long)MT * CTL * 256;
  long long w = 7 * NT + (long long)K * NP + 3 * MS + (NT + MS + 1) / 2 + 64;   // + the certificate's dual snapshot (floats)
  if (accel > 0) w += MS + 3 * (NT + MS) + (2LL * accel * (NT + MS) * 4 + 7) / 8;   // zhr; u, f, g; the float rings
  w += long_flat_doubles(NP);   // the compressed bounds of the row items (at the END of the workspace)
  return w;
}

// CTL: column registers of a row item (>= ceil(Tm / 16)); NWV: waves per problem
// LDSR: the iterates (x, z1, y1, q, lb, ub, r0 / zh and the site-row state) live in LDS instead of the workspace -- for
// problems small enough (N <= 64, horizon <= 32: 7 x 16 KB + site rows + e^, h^ <= 156 KB); only the Anderson vectors,
// the certificate's snapshot and the session multipliers stay in the workspace.
// RZL: only the r0 / zh array lives in LDS (it is the array with the most passes per iteration: written and read by
// both phases, 5 of 15) -- for problems whose one array fits next to e^, h^ (54 x 144: 72 + 18 KB).
// XSL (with RZL): x too (3 passes) -- horizons up to 96 on a 64-EVSE site.
template <int CTL, int MT, int NWV, bool LDSR = false, bool RZL = false, bool XSL = false>
__global__ __launch_bounds__(NWV * 64, 1) void admm_long_kernel(const StreamArgs SA_kernarg) {
  using M = Mfma<double>;
  using vec4 = M::vec4;
  typedef double real;
  constexpr int AMX = kLongAccelMax;
  __shared__ real SC[NWV * 8 + 8];
  __shared__ real AaRedS[NWV * (AMX + 2)];
  __shared__ real AaHS[NWV * (AMX * AMX + AMX)];
  __shared__ real RowLam[16 * MT], RowLim[16 * MT], RowDj[16 * MT];   // per padded site row: eigenvalue, limit, rho / (a + rho lam)
  __shared__ int RowTy[16 * MT];
  __shared__ unsigned char RowFlat[kLongFlatRows];   // per register row (4 EVSEs x the horizon): its bounds compress
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* EH = reinterpret_cast<real*>(smem_raw);   // e^ [MT][CTL][4][64]
  real* HH = EH + MT * CTL * 256;                 // h^ (start: Ghat z1)

  // passes: pass 0 as the options state it, then cold fixed-penalty retries of a stalled problem (retry_wanted,
  // acn_qp_tiled.hpp).  The WHOLE body is the pass, with the argument block read through a per-pass opaque pointer to
  // the kernarg segment: no argument is kept alive across the solver loop for the next pass's sake.
  __shared__ int q_slot;
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
  // Block id through v_readfirstlane (as in the other kernels): the workspace descriptor and every uniform offset then
  // live in scalar registers -- 26 v_readfirstlane in the 54 x 144 instantiation instead of 1,976, 78 -> 63 ms.  That
  // form exposed a hardware hazard the compiler does not cover (see st2 below), which is why round 2 shipped without it.
#if !defined(ACNQP_LONG_IDS_OPAQUE) || ACNQP_LONG_IDS_OPAQUE
  int b_ = q_pos; int tid = threadIdx.x;
  asm volatile("" : "+v"(b_)); asm volatile("" : "+v"(tid));
  const int wg_ = __builtin_amdgcn_readfirstlane(b_);
  const int b = __builtin_amdgcn_readfirstlane(A.order ? A.order[wg_] : wg_);   // the problem this workgroup solves (a uniform value: the load alone would make it a vector register)
#else
  const int b = q_pos, tid = threadIdx.x;   // diagnostic: descriptor in vector registers (round 2's form)
#endif
  if (A.resume) {   // the launch behind the polish kernel (acn_qp_tiled.hpp): what it solved is done; the rest starts over
    if (A.status[b] != kStatusPolish && pass == 0) break;
    if (pass == 0) it_total = A.iters[b];
  }
  const int max_iter_p = pass == 0 ? A.max_iter : min(A.max_iter, A.retry_max_iter);
  const int adapt_p = pass == 0 ? A.adapt_every : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
  int g = lane >> 4, t = lane & 15;
#define RELANE() do { asm volatile("" : "+v"(lane)); g = lane >> 4; t = lane & 15; } while (0)
  const int N = A.N, Tm = A.Tm, NP = A.NP, K = A.K;
  const int NE = NP >> 4;                 // EVSE tiles
  const int nct = (Tm + 15) >> 4;         // column tiles that hold periods (<= CTL); the rest is padding, all zero
  const int n_site = MT * nct, n_tile = n_site + NE * nct;   // tile items: site tiles first, then EVSE tiles
  const long long NT = (long long)NE * CTL * 256;
  // (the workspace belongs to the workgroup slot when the launch runs off the queue: acn_qp_stream.hpp)
  int ws_slot_ = (int)blockIdx.x;   // (opaque per pass: acn_qp_stream.hpp)
  asm volatile("" : "+v"(ws_slot_));
  ws_slot_ = __builtin_amdgcn_readfirstlane(ws_slot_);
  real* W0 = SA.work + (size_t)(A.ws_by_slot ? ws_slot_ : b) * SA.ws_per_problem;
  const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(W0, 0, (int)(SA.ws_per_problem * 8), 0x00020000);
  const unsigned NT8 = (unsigned)NT * 8u;
  const unsigned MS8 = (unsigned)(MT * CTL * 256) * 8u;
  // RZ: r0 (written by phase 2, read by phase 1) / zh (the other way round); Z2, Y2, GX: site-row state in
  // tile-fragment order [MT][CTL][4][64].  Same workspace layout either way (the LDS-resident variant leaves its
  // share of it unused).
  typedef typename std::conditional<LDSR, real*, WsArr64>::type StArr;
  typedef typename std::conditional<LDSR || RZL, real*, WsArr64>::type RzArr;
  typedef typename std::conditional<LDSR || XSL, real*, WsArr64>::type XsArr;
  static_assert(!XSL || RZL, "x joins r0 / zh in LDS, never alone");
  StArr Z1s, Y1s, Qs, LBs, UBs, Z2, Y2, GX;
  RzArr RZ;
  XsArr Xs;
  const unsigned z2off = 7 * NT8 + (unsigned)(K * NP) * 8u;
  if constexpr (LDSR) {
    real* S0 = HH + MT * CTL * 256;
    Xs = S0; Z1s = S0 + NT; Y1s = S0 + 2 * NT; Qs = S0 + 3 * NT; LBs = S0 + 4 * NT; UBs = S0 + 5 * NT; RZ = S0 + 6 * NT;
    Z2 = S0 + 7 * NT; Y2 = Z2 + MT * CTL * 256; GX = Y2 + MT * CTL * 256;
  } else {
    if constexpr (XSL) Xs = HH + MT * CTL * 256 + NT; else Xs = WsArr64{0};
    Z1s = WsArr64{NT8}; Y1s = WsArr64{2 * NT8}; Qs = WsArr64{3 * NT8}; LBs = WsArr64{4 * NT8};
    UBs = WsArr64{5 * NT8};
    if constexpr (RZL) RZ = HH + MT * CTL * 256; else RZ = WsArr64{6 * NT8};
    Z2 = WsArr64{z2off}; Y2 = WsArr64{z2off + MS8}; GX = WsArr64{z2off + 2 * MS8};
  }
  real* MU = W0 + 7 * NT;                 // [K][NP] (indexed per lane: a plain pointer)
  // Anderson acceleration (acn_qp_tiled.hpp, oracle/admm_port.c): pre-projection site rows of an event iteration, the
  // previous event's u / f / g over the EVSE part [0, NT) and the site part [NT, NT + MS), the dF / dG rings (floats)
  const int aa_m = min(max(A.accel_mem, 0), AMX);
  const unsigned MS = (unsigned)(MT * CTL * 256), DU = (unsigned)NT + MS;
  const WsArr32 Y1P{z2off + 3 * MS8};     // duals at the previous check (infeasibility certificate)
  const WsArr32 Y2P{Y1P.off + (unsigned)NT * 4u};
  const WsArr64 ZHR{z2off + 3 * MS8 + ((DU + 1) / 2) * 8u};
  const WsArr64 UP{ZHR.off + MS8}, FP{UP.off + DU * 8u}, GP{FP.off + DU * 8u};
  const WsArr32 HF{GP.off + DU * 8u};
  const WsArr32 HG{HF.off + (unsigned)aa_m * DU * 4u};
  // Flat items.  The rate bounds of a session are two constants inside its window and zero outside (aco.py:45-73: min /
  // max rates of the session, clipped to the EVSE's), so per lane the CTL (l, u) pairs of a row nearly always take ONE
  // non-zero value: the init phase checks exactly that, lane by lane, and keeps (l, u) + a CTL-bit mask; phase 2 then
  // rebuilds the bounds of a row item from 40 bytes per lane instead of streaming 2 x 2 x CTL x 8 -- bit for bit the
  // same numbers.  An item with a lane that does not fit (per-period max_rates) streams its bounds as before.
  constexpr bool kFlat = !LDSR && CTL <= 31 && ACNQP_LONG_FLAT_BOUNDS;   // (the period bits of a row fit one word)
  const bool flat_on = kFlat && 4 * NE <= kLongFlatRows;
  const unsigned cblu = (unsigned)(SA.ws_per_problem - long_flat_doubles(NP)) * 8u;   // [item][row of the pair][lane] (l, u)
  const unsigned cbm = cblu + (unsigned)NE * 4096u;                                   // [item][lane] two masks
  real* AaH = AaHS + wave * (AMX * AMX + AMX);   // this wave's copy of (H, b): every wave runs the small solve itself
  const real* FG = static_cast<const real*>(A.fragG2);   // pair order: a 16-byte load per lane fetches two k-slices
  const real* FQ = static_cast<const real*>(A.fragQ2);
  const real* Gm = static_cast<const real*>(A.G);
  const real* Lm = static_cast<const real*>(A.lam);
  const real* RL = static_cast<const real*>(A.rowlim);
  const bool eq = A.s_eq[b] != 0;
  const real sigma = A.sigma, alpha = A.alpha;
  const real lfb = A.lf ? A.lf[b] / (A.flat_scale * A.flat_scale) : 0.0;
  // demand charge (acn_qp_tiled.hpp): prox of dc * max(max_t z_t, floor) over the WHOLE horizon of the "max" site row
  const real dcb = A.dc ? A.dc[b] / A.max_scale : 0.0;
  const real dfl = A.dfloor ? A.dfloor[b] * A.max_scale : 0.0;
  int jdc = -1;   // padded index of that row (uniform)
  if (dcb > 0.0)
    for (int j = 0; j < 16 * MT; ++j) jdc = A.rowtype[j] == kRowMax ? j : jdc;
  const bool dc_on = jdc >= 0;

  // Layout of every state array (workspace or LDS): 16 x 16 tiles in MFMA fragment order, the four accumulator
  // registers of a lane stored as two adjacent PAIRS -- element (tile, r, lane) at (tile * 2 + r / 2) * 128 + lane * 2 +
  // r % 2 -- so that one 16-byte access per lane moves two registers: the CU's address unit prices a 64-lane access
  // at ~16 cycles whatever its width (tools/micro/rt_latency.hip), and an iteration is bound by that count.
  // Addressing: (array + wave-uniform offset)[lane] -- a scalar base / buffer descriptor and one 32-bit lane offset per
  // instruction instead of a 64-bit per-lane address computed for each of them.  fidx / sidx2 are the uniform parts
  // (they include r % 2); roff(r) is the offset of register r inside its tile.
  auto roff = [&](int r) -> unsigned { return (unsigned)((r >> 1) * 128 + (r & 1)); };
  auto fidx = [&](int e, int c, int r) -> unsigned { return (unsigned)((e * CTL + c) * 256) + roff(r); };
  auto sidx2 = [&](int m, int c, int r) -> unsigned { return (unsigned)((m * CTL + c) * 256) + roff(r); };
  // one element (8 / 4 bytes) of a state array; the lane's byte offset is formed in 32 bits and zero-extended: the
  // shape the scalar-base addressing mode takes
  auto at = [&](auto base, unsigned uo) -> decltype(auto) {
    typedef decltype(base) B;
    if constexpr (std::is_same<B, WsArr64>::value) return WsRef64{wsr, (unsigned)lane * 16u, base.off + uo * 8u};
    else if constexpr (std::is_same<B, WsArr32>::value) return WsRef32{wsr, (unsigned)lane * 8u, base.off + uo * 4u};
    else {   // LDS-resident state
      typedef typename std::remove_pointer<B>::type elem_t;
      typedef typename std::conditional<std::is_const<elem_t>::value, const char, char>::type byte_t;
      const unsigned lo = (unsigned)lane * 2u * (unsigned)sizeof(elem_t);
      return (*reinterpret_cast<B>(reinterpret_cast<byte_t*>(base + uo) + lo));
    }
  };
  // two adjacent registers (2 rp, 2 rp + 1; uo = fidx / sidx2 of the even one) in one 16-byte access (float arrays:
  // 8-byte, widened to / rounded from double exactly as the single-element accessor does)
  // kill (wave-uniform, 0 or kWsKill): a workspace access whose lane offset carries kWsKill is out of the descriptor's
  // range -- the load returns zeros and the store is dropped, neither touches memory.  The tile phases pass it for a
  // register pair whose eight EVSEs are all padding (their state is zero and stays zero); LDS-resident arrays ignore it.
  auto ld2 = [&](auto base, unsigned uo, real& a, real& b, unsigned kill = 0u) __attribute__((always_inline)) {
    typedef decltype(base) B;
    if constexpr (std::is_same<B, WsArr64>::value) {
      const ws_d2 d = __builtin_bit_cast(ws_d2, __builtin_amdgcn_raw_buffer_load_b128(wsr, ((unsigned)lane * 16u) | kill, base.off + uo * 8u, 0));
      a = d.x; b = d.y;
    } else if constexpr (std::is_same<B, WsArr32>::value) {
      const ws_f2 d = __builtin_bit_cast(ws_f2, __builtin_amdgcn_raw_buffer_load_b64(wsr, ((unsigned)lane * 8u) | kill, base.off + uo * 4u, 0));
      a = d.x; b = d.y;
    } else {   // LDS-resident state (doubles)
      const ws_d2 d = *reinterpret_cast<const ws_d2*>(reinterpret_cast<const char*>(base + uo) + (unsigned)lane * 16u);
      a = d.x; b = d.y;
    }
  };
  auto st2 = [&](auto base, unsigned uo, real a, real b, unsigned kill = 0u) __attribute__((always_inline)) {
    typedef decltype(base) B;
    if constexpr (std::is_same<B, WsArr64>::value) {
      const ws_d2 d = {a, b};
      const ws_v4u q4 = __builtin_bit_cast(ws_v4u, d);
      __builtin_amdgcn_raw_buffer_store_b128(q4, wsr, ((unsigned)lane * 16u) | kill, base.off + uo * 8u, 0);
      // gfx950 store-data hazard: a buffer store of more than 64 bits per lane reads its data registers over several
      // cycles, and a VALU write to them in the next two issue slots changes what the last lanes (12..15 of each DPP
      // row) store.  LLVM inserts the wait states for such stores EXCEPT when soffset is a register
      // (GCNHazardRecognizer::createsVALUHazard) -- exactly this instruction, and on MI355X the exemption does not hold:
      // the new r0 of column tile c was stored with the value of tile c + 1 in those lanes, differently from run to run
      // (tools/gpu_long_race.py located it).  The asm keeps the store's own register tuple live across
      // three wait states; tools/check_store_hazard.py scans the ISA of every kernel for the pattern at build time.
#if !defined(ACNQP_LONG_STORE_NOP) || ACNQP_LONG_STORE_NOP
      asm volatile("s_nop 2" ::"v"(q4));   // the 128-bit tuple the store reads, not the doubles it was built from
#endif
    } else if constexpr (std::is_same<B, WsArr32>::value) {
      const ws_f2 d = {(float)a, (float)b};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ws_v2u, d), wsr, ((unsigned)lane * 8u) | kill, base.off + uo * 4u, 0);
    } else {
      const ws_d2 d = {a, b};
      *reinterpret_cast<ws_d2*>(reinterpret_cast<char*>(base + uo) + (unsigned)lane * 16u) = d;
    }
  };
  // the four registers of a tile (uo = fidx / sidx2 of register 0)
  // (kill2: the kill flag of the SECOND pair, registers 2 and 3 = rows 8 .. 15 of the tile)
  auto ld4 = [&](auto base, unsigned uo, real (&v)[4], unsigned kill2 = 0u) __attribute__((always_inline)) {
    ld2(base, uo, v[0], v[1]);
    ld2(base, uo + 128u, v[2], v[3], kill2);
  };
  auto st4 = [&](auto base, unsigned uo, const real (&v)[4], unsigned kill2 = 0u) __attribute__((always_inline)) {
    st2(base, uo, v[0], v[1]);
    st2(base, uo + 128u, v[2], v[3], kill2);
  };
  // rows 8 .. 15 of EVSE tile e are all padding (54 EVSEs: rows 56 .. 63 of tile 3): their pair is not streamed
  auto pad2 = [&](int e) -> unsigned { return 16 * e + 8 >= N ? kWsKill : 0u; };
  // the four k-slices of one 4 x 64 fragment block of the site matrices (pair order: two 16-byte loads)
  auto ldf4 = [&](const real* blk, real (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const ws_d2 d = *reinterpret_cast<const ws_d2*>(reinterpret_cast<const char*>(blk + 128 * h) + (unsigned)lane * 16u);
      v[2 * h] = d.x; v[2 * h + 1] = d.y;
    }
  };

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
    for (int c = 0; c < CTL; ++c) {   // every column register, padding included (zeros): the row passes never branch
      const int tt = 16 * c + t;
      const bool ok = ev < N && tt < Tm;
      const size_t idx = ((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0);
      const real l = ok ? A.lb[idx] : 0.0;
      real u = ok ? A.ub[idx] : 0.0;
      const real q = ok ? A.q[idx] : 0.0;
      if (u < l) u = l;
      lbv[c] = l; ubv[c] = u;
      const unsigned i = fidx(e, c, r);
      at(LBs, i) = l; at(UBs, i) = u; at(Qs, i) = q;
      qn = fmax(qn, fabs(q)); um = fmax(um, u);
    }
    if constexpr (kFlat) {
      if (flat_on) {   // (block-uniform)
        real L = lbv[0], U = ubv[0];
#pragma unroll
        for (int c = 1; c < CTL; ++c) {   // the pair with the largest u (then l): the candidate "on" value of this lane
          const bool up = ubv[c] > U || (ubv[c] == U && lbv[c] > L);
          L = up ? lbv[c] : L; U = up ? ubv[c] : U;
        }
        auto bits = [](real v) -> long long { return __builtin_bit_cast(long long, v); };
        unsigned mk = 0; bool okl = true;
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          // (bit patterns, not values: the rebuilt bounds are the stored ones to the last bit, signed zeros included)
          const bool on = bits(lbv[c]) == bits(L) && bits(ubv[c]) == bits(U), off = bits(lbv[c]) == 0 && bits(ubv[c]) == 0;
          okl = okl && (on || off);
          mk |= on && !off ? 1u << c : 0u;
        }
        const int pi = ri >> 1, hr = ri & 1;
        const unsigned o = cblu + (unsigned)(pi * 2 + hr) * 1024u;
        WsRef64{wsr, (unsigned)lane * 16u, o} = L;
        WsRef64{wsr, (unsigned)lane * 16u, o + 8u} = U;
        __builtin_amdgcn_raw_buffer_store_b32(mk, wsr, (unsigned)lane * 8u, cbm + (unsigned)pi * 512u + (unsigned)hr * 4u, 0);
        const bool okrow = __builtin_amdgcn_ballot_w64(!okl) == 0;
        if (lane == 0) RowFlat[ri] = okrow ? 1 : 0;   // read after the barriers of the block maximum below
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
    pd = effective_pdiag<real>(pd_user, A.reg_rel, qnorm, f[1], A.horizon[b], lfb > 0.0 || dcb > 0.0);
    if (f[2] > 0) {
      for (size_t k = tid; k < (size_t)N * Tm; k += NWV * 64) A.x[(size_t)b * N * Tm + k] = 0;
      if (A.y_out && !A.y_for_polish_only)
        for (size_t k = tid; k < (size_t)A.Mg * Tm; k += NWV * 64) A.y_out[(size_t)b * A.Mg * Tm + k] = 0;
      if (tid == 0) { A.status[b] = 4; A.iters[b] = 0; A.pri[b] = M::big; A.dua[b] = M::big; A.obj[b] = 0; }
      break;   // (block-uniform) out of the pass loop: the next problem of the queue
    }
  }

  real rho = A.rho0;
  if (pass > 0) {   // fixed penalty retry_rho * 4^(pass - 1)
    rho = A.retry_rho;
    for (int k = 1; k < pass; ++k) rho *= 4.0;
  }

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


  // the site's row constants -> LDS
  for (int j = tid; j < 16 * MT; j += NWV * 64) { RowLam[j] = Lm[j]; RowLim[j] = RL[j]; RowTy[j] = A.rowtype[j]; }
  auto set_dj = [&]() {   // rho / (a + rho lam_j): after every rho change (a barrier follows every call)
    const real a_ = sigma + pd + rho;
    for (int j = tid; j < 16 * MT; j += NWV * 64) RowDj[j] = rho / (a_ + rho * Lm[j]);
  };
  set_dj();

  // tile item q -> (site tile mo, c) for q < n_site, (EVSE tile e, c) after; column fastest.  The quotient of the
  // (uniform) division comes out of the vector ALU: readfirstlane puts it back into a scalar register, so that the
  // tile addresses stay scalar.
  // ---- projection of one site-row tile onto C from its pre-projection point: z2, y2, the tile's residual terms ----
  real sv0 = 0, sv2 = 0;   // |G x - z2|_inf, max(|G x|, |z2|): the site-row share of the residuals, per iteration
  auto site_project = [&](int mo, int c, const real (&zhr)[4], const real (&gxn)[4], real quad) __attribute__((always_inline)) {
    const int tt = 16 * c + t;
    real pk = M::big;
    if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
    int ty[4];
    real lim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int j = 16 * mo + M::rowof(g, r); ty[r] = RowTy[j]; lim[r] = RowLim[j]; }
    real scl[2] = {1.0, 1.0};
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
      if (ty[2 * pr] == kRowSocRe) {
        const real re = zhr[2 * pr], im = zhr[2 * pr + 1];
        const real n2 = re * re + im * im;
        if (n2 > lim[2 * pr] * lim[2 * pr]) scl[pr] = lim[2 * pr] * rsqrt_nr(n2);
      }
    real z2n[4], y2n[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      real zn = zhr[r];
      if (ty[r] == kRowBox) zn = fmin(zn, lim[r]);
      else if (ty[r] == kRowPeak) zn = fmin(zn, pk);
      else if (ty[r] == kRowQuad) zn = zn * quad;
      else if (ty[r] == kRowSocRe || ty[r] == kRowSocIm) zn = zn * scl[r >> 1];
      // kRowMax: zn = zhr here; the horizon-wide prox (dc_row below) follows once every column tile is through
      y2n[r] = rho * (zhr[r] - zn);
      z2n[r] = zn;
      if (!(dc_on && ty[r] == kRowMax)) {
        sv0 = fmax(sv0, fabs(gxn[r] - zn));
        sv2 = fmax(sv2, fmax(fabs(gxn[r]), fabs(zn)));
      }
    }
    st4(Y2, sidx2(mo, c, 0), y2n);
    st4(Z2, sidx2(mo, c, 0), z2n);
  };
  // ---- demand charge: z_t = min(zh_t, max(tau, floor)), tau = root of sum_t (zh_t - tau)+ = dc / rho (Newton on a convex
  // piecewise-linear function), over the whole horizon of the "max" row: one wave, after every site tile has stored
  // its pre-projection point of the row in Z2.  The row's periods are the 16 lanes of ONE lane group (g = jdc & 3) x the
  // column tiles; the other three lane groups run along on their own rows and store nothing.
  auto dc_row = [&]() __attribute__((always_inline)) {
    const int mo = jdc >> 4, rr = jdc & 15, gd = rr & 3, rd = rr >> 2;   // rowof(g, r) = g + 4 r
    real zv[CTL], gxv[CTL];
#pragma unroll
    for (int c = 0; c < CTL; ++c) {
      zv[c] = 0; gxv[c] = 0;
      if (c < nct) { zv[c] = at(Z2, sidx2(mo, c, rd)); gxv[c] = at(GX, sidx2(mo, c, rd)); }
    }
    const real cw = dcb / rho;
    real vmax_l = -M::big;
#pragma unroll
    for (int c = 0; c < CTL; ++c) vmax_l = (16 * c + t < Tm) ? fmax(vmax_l, zv[c]) : vmax_l;
    const real vmax = row_max<real>(vmax_l);
    real tau = vmax - cw;
    bool need = g == gd;
    int guard = 0;
    while (__any(need)) {
      ++guard;
      real sl = 0, nl = 0;
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
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
      for (int c = 0; c < CTL; ++c)
        if (c < nct) {
          const real zn = (16 * c + t < Tm) ? fmin(zv[c], lev) : zv[c];
          at(Y2, sidx2(mo, c, rd)) = rho * (zv[c] - zn);
          at(Z2, sidx2(mo, c, rd)) = zn;
          sv0 = fmax(sv0, fabs(gxv[c] - zn));
          sv2 = fmax(sv2, fmax(fabs(gxv[c]), fabs(zn)));
        }
    }
  };
  // P tile = sum over EVSE tiles of Ghat[mo, e] v[e, c], v = the RZ array (r0, or z1 during the start), in tile order,
  // two tiles' operands per batch of loads
  auto site_p = [&](int mo, int c, const real* fgb) -> vec4 {
    vec4 p = {0, 0, 0, 0};
#pragma unroll 1
    for (int e = 0; e < NE; e += 2) {
      const int e1 = e + 1 < NE ? e + 1 : e;
      const real on1 = e + 1 < NE ? 1.0 : 0.0;
      real b0[4], a0[4], b1[4], a1[4];
      const real* f0 = fgb + (size_t)(e * MT + mo) * 2 * 4 * 64;
      const real* f1 = fgb + (size_t)(e1 * MT + mo) * 2 * 4 * 64;
      ld4(RZ, fidx(e, c, 0), b0);
      ld4(RZ, fidx(e1, c, 0), b1);
      ldf4(f0, a0); ldf4(f1, a1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) p = M::mma(a0[s], b0[s], p);
#pragma unroll
      for (int s = 0; s < 4; ++s) p = M::mma(a1[s], b1[s] * on1, p);
    }
    return p;
  };
  // r0 and (if accelerated) u of one EVSE tile from the stored state: start, rho change
  auto reset_evse = [&](int e, int c) __attribute__((always_inline)) {
    const real ir = 1.0 / rho;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned i = fidx(e, c, r);
      const real z = at(Z1s, i), y = at(Y1s, i);
      at(RZ, i) = sigma * at(Xs, i) - at(Qs, i) + rho * z - y;
      if (aa_m > 0) at(UP, i) = z + y * ir;
    }
  };
  auto reset_site = [&](int mo, int c) __attribute__((always_inline)) {
    const real ir = 1.0 / rho;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(mo, c, r); at(UP, (unsigned)NT + i) = at(Z2, i) + at(Y2, i) * ir; }
  };

  // ---- start (see acn_qp_tiled.hpp): z1 = Proj_B(-kStartGain q), x = z1, y1 = -(q + pd z1); z2 = G z1 = Q (Ghat z1);
  // warm (optional): z1 = Proj_B(warm_x), y2 = warm_y, y1 = -(q + pd z1 + G' y2) -------------------------------------
#ifdef ACNQP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  const bool warm = pass == 0 && A.warm_x != nullptr && A.warm_y != nullptr;
#pragma unroll 1
  for (int ri = wave; ri < 4 * NE; ri += NWV) {
    RELANE();
    const int e = ri >> 2, r = ri & 3;
    real zs[CTL], lbv[CTL], ubv[CTL], z1[CTL];
#pragma unroll
    for (int c = 0; c < CTL; ++c) {
      const unsigned i = fidx(e, c, r);
      lbv[c] = at(LBs, i); ubv[c] = at(UBs, i);
      zs[c] = -kStartGain * at(Qs, i);
      if (warm) {
        const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
        const bool ok = ev < N && tt < Tm;
        zs[c] = ok ? A.warm_x[((size_t)b * N + (ok ? ev : 0)) * Tm + (ok ? tt : 0)] : 0.0;
      }
    }
    project_row(e, r, zs, lbv, ubv, z1, true);
#pragma unroll
    for (int c = 0; c < CTL; ++c) {
      const unsigned i = fidx(e, c, r);
      at(Xs, i) = z1[c]; at(Z1s, i) = z1[c]; at(Y1s, i) = -(at(Qs, i) + pd * z1[c]); at(RZ, i) = z1[c];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int q = wave; q < n_site; q += NWV) {   // Ghat z1 -> HH
    RELANE();
    const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
    const vec4 p = site_p(mo, c, FG);
#pragma unroll
    for (int r = 0; r < 4; ++r) at(HH, sidx2(mo, c, r)) = p[r];
  }
  __syncthreads();
#pragma unroll 1
  for (int q = wave; q < n_site; q += NWV) {   // z2 = G z1 = Q (Ghat z1), y2
    RELANE();
    const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
    vec4 zt = {0, 0, 0, 0};
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      real fq[4], hv[4];
      ldf4(FQ + ((mo * MT + mi) * 2 + 1) * 256, fq);
      ld4(HH, sidx2(mi, c, 0), hv);
#pragma unroll
      for (int s = 0; s < 4; ++s) zt = M::mma(fq[s], hv[s], zt);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned i = sidx2(mo, c, r);
      real yv = 0;
      if (warm) {
        const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
        const int ja = A.rowabi[j];
        if (ja >= 0 && tt < Tm) yv = A.warm_y[((size_t)b * A.Mg + ja) * Tm + tt] / static_cast<const real*>(A.rowscale)[j];
      }
      at(Z2, i) = zt[r]; at(GX, i) = zt[r]; at(Y2, i) = yv;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int q = wave; q < n_tile; q += NWV) {
    RELANE();
    if (q < n_site) {
      const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
      if (aa_m > 0) reset_site(mo, c);
    } else {
      const int qe = q - n_site;
      const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
      if (warm) {   // y1 = -(q + pd z1 + G' y2): the row items stored the G' y2 = 0 version
        vec4 gty = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], at(Y2, sidx2(m, c, s)), gty);
#pragma unroll
        for (int r = 0; r < 4; ++r) at(Y1s, fidx(e, c, r)) -= gty[r];
      }
      reset_evse(e, c);
    }
  }
  __syncthreads();

  int status = 2, it = 0, n_adapt = 0, best_it = 0;
  real best_score = M::big;
  real pri = M::big, dua = M::big;
  bool done = false, have_prev = false;
  // Anderson state (block-uniform scalars, every wave keeps its own identical copy)
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  if (aa_m > 0)
    for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
#pragma unroll 1
  while (!done) {
    ++it;
    const real a = sigma + pd + rho, inv_a = 1.0 / a, inv_rho = 1.0 / rho;
    const real quad = rho / (rho + lfb);
    const bool check = (it % A.check_every == 0) || it >= max_iter_p;
    // an offset the compiler cannot see through keeps the loads of loop-invariant site data inside the loop (L1 / L2
    // hits) instead of pinning registers across it
    unsigned zoff = 0;
    asm volatile("" : "+s"(zoff));
    const real* FQi = FQ + zoff;
    const real* FGi = FG + zoff;
    sv0 = 0; sv2 = 0;
    const bool ev_it = aa_m > 0 && it % kAaPeriod == 0;   // Anderson event: the site rows are projected after it
    // ================= phase 1a: site tiles -> e^, h^ ============================================================
#pragma unroll 1
    for (int q = wave; q < n_site; q += NWV) {
      RELANE();
      const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
      real bz[MT][4], fq0[MT][4];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        real z2t[4], y2t[4];
        ld4(Z2, sidx2(mi, c, 0), z2t);
        ld4(Y2, sidx2(mi, c, 0), y2t);
        ldf4(FQi + ((mo * MT + mi) * 2 + 0) * 256, fq0[mi]);
#pragma unroll
        for (int s = 0; s < 4; ++s) bz[mi][s] = rho * z2t[s] - y2t[s];
      }
      const vec4 p = site_p(mo, c, FGi);
      vec4 wh = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) wh = M::mma(fq0[mi][s], bz[mi][s], wh);
      real eo[4], ho[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mo + M::rowof(g, r);
        const real lj = RowLam[j];
        eo[r] = wh[r] - RowDj[j] * (p[r] + lj * wh[r]);
        ho[r] = (p[r] + lj * eo[r]) * inv_a;
      }
      st4(EH, sidx2(mo, c, 0), eo);
      st4(HH, sidx2(mo, c, 0), ho);
    }
    STAMP(0);   // 1a
    __syncthreads();
    STAMP(1);   // barrier
    // ================= phase 1b: site tiles (G x~, projection onto C) and EVSE tiles (x~, zh) =======================
#pragma unroll 1
    for (int q = wave; q < n_tile; q += NWV) {
      RELANE();
      if (q < n_site) {
        const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
        real fq1[MT][4], hv[MT][4], gxv[4], z2v[4], y2v[4];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          ld4(HH, sidx2(mi, c, 0), hv[mi]);
          ldf4(FQi + ((mo * MT + mi) * 2 + 1) * 256, fq1[mi]);
        }
        ld4(GX, sidx2(mo, c, 0), gxv); ld4(Z2, sidx2(mo, c, 0), z2v); ld4(Y2, sidx2(mo, c, 0), y2v);
        __builtin_amdgcn_sched_barrier(0);
        vec4 zt = {0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) zt = M::mma(fq1[mi][s], hv[mi][s], zt);
        real zhr[4], gxn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gxn[r] = alpha * zt[r] + (1.0 - alpha) * gxv[r];
          zhr[r] = alpha * zt[r] + (1.0 - alpha) * z2v[r] + y2v[r] * inv_rho;
        }
        st4(GX, sidx2(mo, c, 0), gxn);
        if (ev_it) {
          st4(ZHR, sidx2(mo, c, 0), zhr);
        } else {
          site_project(mo, c, zhr, gxn, quad);
        }
      } else {
        const int qe = q - n_site;
        const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
        const real* fg = FGi + (size_t)e * MT * 2 * 4 * 64;
        const unsigned kl = pad2(e);
        real rv[4], xv[4], zv[4], yv[4], fx[MT][4], ev4[MT][4];
        ld4(RZ, fidx(e, c, 0), rv, kl); ld4(Xs, fidx(e, c, 0), xv, kl); ld4(Z1s, fidx(e, c, 0), zv, kl); ld4(Y1s, fidx(e, c, 0), yv, kl);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          ld4(EH, sidx2(m, c, 0), ev4[m]);
          ldf4(fg + (m * 2 + 1) * 256, fx[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
        vec4 acc = {rv[0], rv[1], rv[2], rv[3]};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = M::mma(fx[m][s], ev4[m][s], acc);
        real zo4[4], xo4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real xn = acc[r] * inv_a;
          zo4[r] = alpha * xn + (1.0 - alpha) * zv[r] + yv[r] * inv_rho;
          xo4[r] = alpha * xn + (1.0 - alpha) * xv[r];
        }
        st4(RZ, fidx(e, c, 0), zo4, kl);
        st4(Xs, fidx(e, c, 0), xo4, kl);
      }
    }
    STAMP(2);   // 1b
    if (ev_it) {
      // ---- Anderson event (type II, acn_qp_tiled.hpp / oracle/admm_port.c): u = (zh, zhr) is the state of the
      // fixed-point map.  Tile items again: the wave that wrote a tile's zh / zhr reads it back.
      const bool col = aa_have_prev;
      const int slot = aa_head;
      real d[AMX + 2];
#pragma unroll
      for (int j = 0; j < AMX + 2; ++j) d[j] = 0;
      // four registers of one tile: g = zsrc[zo + 64 r], state index uo + 64 r
      auto aa_tile = [&](auto zsrc, unsigned zo, unsigned uo, unsigned kl) __attribute__((always_inline)) {
        real gv[4], uv[4], fpv[4], gpv[4], hv[AMX][4], cqv[4], cgv[4], fv[4];
        ld4(zsrc, zo, gv, kl); ld4(UP, uo, uv, kl); ld4(FP, uo, fpv, kl); ld4(GP, uo, gpv, kl);
        // every ring column is requested, live or not (a branch per column puts each load in a basic block of its own:
        // five dependent memory round trips per tile instead of one); a dead column's data is replaced by zeros
#pragma unroll
        for (int j = 0; j < AMX; ++j) ld4(HF + (size_t)(j < aa_m ? j : aa_m - 1) * DU, uo, hv[j], kl);
#pragma unroll
        for (int j = 0; j < AMX; ++j) {
          const bool live = ((aa_valid >> j) & 1u) && j != slot;   // uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[j][r] = live ? hv[j][r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real f = gv[r] - uv[r];
          d[AMX + 1] += f * f;
          const float cq = (float)(f - fpv[r]);
          cqv[r] = cq; cgv[r] = (float)(gv[r] - gpv[r]); fv[r] = f;
#pragma unroll
          for (int j = 0; j < AMX; ++j) d[j] += (real)cq * (j == slot ? (real)cq : hv[j][r]);
          d[AMX] += (real)cq * f;
        }
        if (col) { st4(HF + (size_t)slot * DU, uo, cqv, kl); st4(HG + (size_t)slot * DU, uo, cgv, kl); }
        st4(FP, uo, fv, kl); st4(GP, uo, gv, kl);
      };
#pragma unroll 1
      for (int q = wave; q < n_tile; q += NWV) {
        RELANE();
        if (q < n_site) {
          const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
          aa_tile(ZHR, sidx2(mo, c, 0), (unsigned)NT + sidx2(mo, c, 0), 0u);
        } else {
          const int qe = q - n_site;
          const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
          aa_tile(RZ, fidx(e, c, 0), fidx(e, c, 0), pad2(e));
        }
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
      // ---- apply: u = g - sum_j gamma_j dG_j; the site rows are projected from their (extrapolated) point -------
      auto aa_apply = [&](auto zdst, unsigned zo, unsigned uo, real (&out)[4], unsigned kl) __attribute__((always_inline)) {
        real hv[AMX][4];
        ld4(zdst, zo, out, kl);
        if (ext) {
#pragma unroll
          for (int j = 0; j < AMX; ++j) ld4(HG + (size_t)(j < aa_m ? j : aa_m - 1) * DU, uo, hv[j], kl);   // one batch, as in aa_tile
#pragma unroll
          for (int j = 0; j < AMX; ++j) {
            const bool live = (aa_valid >> j) & 1u;   // uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] -= gam[j] * (live ? hv[j][r] : 0.0);
          }
        }
        st4(UP, uo, out, kl);
      };
#pragma unroll 1
      for (int q = wave; q < n_tile; q += NWV) {
        RELANE();
        if (q < n_site) {
          const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
          real zhr[4], gxn[4];
          aa_apply(ZHR, sidx2(mo, c, 0), (unsigned)NT + sidx2(mo, c, 0), zhr, 0u);
          ld4(GX, sidx2(mo, c, 0), gxn);
          site_project(mo, c, zhr, gxn, quad);
        } else {
          const int qe = q - n_site;
          const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
          real o4[4];
          aa_apply(RZ, fidx(e, c, 0), fidx(e, c, 0), o4, pad2(e));
          if (ext) st4(RZ, fidx(e, c, 0), o4, pad2(e));
        }
      }
    }
    STAMP(3);   // Anderson event (amortised)
    __syncthreads();
    STAMP(4);   // barrier
    // ================= phase 2: row items (and the demand-charge row, by the last wave) ============================
    if (dc_on && wave == NWV - 1) { RELANE(); dc_row(); }
    if constexpr (CTL <= kLongPairRows) {
      // an item = the two rows of a register pair (8 EVSEs x the whole horizon): every access moves 16 bytes per lane
#pragma unroll 1
      for (int pi = wave; pi < 2 * NE; pi += NWV) {
        RELANE();
        const int e = pi >> 1, r0_ = (pi & 1) * 2;
        // an item whose eight EVSEs are all padding (54 EVSEs: rows 56..63, one item in eight) holds zeros in every array
        // and keeps them (lb = ub = q = 0, zero columns of Ghat): it is not streamed at all -- 1/8 of this phase's bytes
        // on the Caltech-shaped site, where the kernel is bandwidth-bound at 2,048 problems (block-uniform per wave)
        if (16 * e + 4 * r0_ >= N) continue;
        real zh2[2][CTL], lb2[2][CTL], ub2[2][CTL], z1p[2][CTL];
        bool flat = false;
        if constexpr (kFlat) flat = flat_on && __builtin_amdgcn_readfirstlane((int)RowFlat[2 * pi] & (int)RowFlat[2 * pi + 1]) != 0;
        if (flat) {
          const ws_d2 lu0 = __builtin_bit_cast(ws_d2, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, cblu + (unsigned)pi * 2048u, 0));
          const ws_d2 lu1 = __builtin_bit_cast(ws_d2, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, cblu + (unsigned)pi * 2048u + 1024u, 0));
          const ws_v2u mk = __builtin_amdgcn_raw_buffer_load_b64(wsr, (unsigned)lane * 8u, cbm + (unsigned)pi * 512u, 0);
#pragma unroll
          for (int c = 0; c < CTL; ++c) {
            ld2(RZ, fidx(e, c, r0_), zh2[0][c], zh2[1][c]);
            const bool on0 = (mk.x >> c) & 1u, on1 = (mk.y >> c) & 1u;
            lb2[0][c] = on0 ? lu0.x : 0.0; ub2[0][c] = on0 ? lu0.y : 0.0;
            lb2[1][c] = on1 ? lu1.x : 0.0; ub2[1][c] = on1 ? lu1.y : 0.0;
          }
        } else {
#pragma unroll
          for (int c = 0; c < CTL; ++c) {   // padding columns hold zeros and stay zero
            const unsigned i = fidx(e, c, r0_);
            ld2(RZ, i, zh2[0][c], zh2[1][c]); ld2(LBs, i, lb2[0][c], lb2[1][c]); ld2(UBs, i, ub2[0][c], ub2[1][c]);
          }
        }
        project_row(e, r0_, zh2[0], lb2[0], ub2[0], z1p[0], false);
        project_row(e, r0_ + 1, zh2[1], lb2[1], ub2[1], z1p[1], false);
        STAMP(5);   // row loads, water-filling
        // x and q of the rows as one batch (into the registers of the bounds), then y1, the new r0 and the stores
#pragma unroll
        for (int c = 0; c < CTL; ++c) { const unsigned i = fidx(e, c, r0_); ld2(Xs, i, lb2[0][c], lb2[1][c]); ld2(Qs, i, ub2[0][c], ub2[1][c]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          const unsigned i = fidx(e, c, r0_);
          const real ya = rho * (zh2[0][c] - z1p[0][c]), yb = rho * (zh2[1][c] - z1p[1][c]);
          st2(Z1s, i, z1p[0][c], z1p[1][c]);
          st2(Y1s, i, ya, yb);
          st2(RZ, i, sigma * lb2[0][c] - ub2[0][c] + rho * z1p[0][c] - ya, sigma * lb2[1][c] - ub2[1][c] + rho * z1p[1][c] - yb);   // the new r0
        }
        STAMP(6);   // y1, new r0
      }
    } else {
      // rows of more than 144 periods: one row per item (two would not fit the registers), 8-byte accesses
#pragma unroll 1
    for (int ri = wave; ri < 4 * NE; ri += NWV) {
      RELANE();
      const int e = ri >> 2, r = ri & 3;
      if (16 * e + 4 * r >= N) continue;   // four padding EVSEs: zeros that stay zeros (see the pair-row loop)
      real zh[CTL], lbv[CTL], ubv[CTL], z1[CTL];
      bool flat = false;
      if constexpr (kFlat) flat = flat_on && __builtin_amdgcn_readfirstlane((int)RowFlat[ri]) != 0;   // this row alone
      if (flat) {   // flat items: (l, u) + the row's period bits instead of 2 x CTL x 8 bytes (same slots as the pair items)
        const ws_d2 lu = __builtin_bit_cast(ws_d2, __builtin_amdgcn_raw_buffer_load_b128(wsr, (unsigned)lane * 16u, cblu + (unsigned)ri * 1024u, 0));
        const unsigned mk = __builtin_amdgcn_raw_buffer_load_b32(wsr, (unsigned)lane * 8u, cbm + (unsigned)(ri >> 1) * 512u + (unsigned)(ri & 1) * 4u, 0);
#pragma unroll
        for (int c = 0; c < CTL; ++c) {
          zh[c] = at(RZ, fidx(e, c, r));
          const bool on = (mk >> c) & 1u;
          lbv[c] = on ? lu.x : 0.0; ubv[c] = on ? lu.y : 0.0;
        }
      } else {
#pragma unroll
        for (int c = 0; c < CTL; ++c) {   // padding columns hold zeros and stay zero
          const unsigned i = fidx(e, c, r);
          zh[c] = at(RZ, i); lbv[c] = at(LBs, i); ubv[c] = at(UBs, i);
        }
      }
      project_row(e, r, zh, lbv, ubv, z1, false);
      STAMP(5);   // row loads, water-filling
      // x and q of the row as one batch (into the registers of the bounds), then y1, the new r0 and the stores
#pragma unroll
      for (int c = 0; c < CTL; ++c) { const unsigned i = fidx(e, c, r); lbv[c] = at(Xs, i); ubv[c] = at(Qs, i); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CTL; ++c) {
        const unsigned i = fidx(e, c, r);
        const real y1n = rho * (zh[c] - z1[c]);
        at(Z1s, i) = z1[c]; at(Y1s, i) = y1n;
        at(RZ, i) = sigma * lbv[c] - ubv[c] + rho * z1[c] - y1n;   // the new r0
      }
      STAMP(6);   // y1, new r0
    }
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);   // barrier
    if (check) {
      // ---- residuals (EVSE tile items; state re-read: L2-hot); (G' y2) by MFMA with the un-rotated site matrix ------
      real v0 = sv0, v1 = 0, v2 = sv2, v4 = 0, v5 = 0;
#pragma unroll 1
      for (int q = wave + n_site; q < n_tile; q += NWV) {
        RELANE();
        const int qe = q - n_site;
        const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
        vec4 gty = {0, 0, 0, 0};
        real xs4[4], qs4[4], ys4[4], zs4[4];
        ld4(Xs, fidx(e, c, 0), xs4); ld4(Qs, fidx(e, c, 0), qs4); ld4(Y1s, fidx(e, c, 0), ys4); ld4(Z1s, fidx(e, c, 0), zs4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          real y2t[4];
          ld4(Y2, sidx2(m, c, 0), y2t);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            gty = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], y2t[s], gty);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real xk = xs4[r], qk = qs4[r], yk = ys4[r], zk = zs4[r];
          v0 = fmax(v0, fabs(xk - zk));
          v1 = fmax(v1, fabs(pd * xk + qk + yk + gty[r]));
          v2 = fmax(v2, fmax(fabs(xk), fabs(zk)));
          v4 = fmax(v4, fabs(pd * xk));
          v5 = fmax(v5, fabs(yk + gty[r]));
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
        for (int q = wave; q < n_tile; q += NWV) {
          RELANE();
          if (q < n_site) {
            const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(mo, c, r); w6[0] = fmax(w6[0], fabs(at(Y2, i) - (real)at(Y2P, i))); }
          } else {
            const int qe = q - n_site;
            const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
            vec4 gtv = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                const unsigned i = sidx2(m, c, s);
                gtv = M::mma(Gm[(size_t)(16 * m + M::rowof(g, s)) * NP + 16 * e + t], at(Y2, i) - (real)at(Y2P, i), gtv);
              }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned i = fidx(e, c, r);
              const real v1_ = at(Y1s, i) - (real)at(Y1P, i);
              w6[0] = fmax(w6[0], fabs(v1_));
              w6[1] = fmax(w6[1], fabs(v1_ + gtv[r]));
            }
          }
        }
        stream_block_max<2, NWV>(w6, SC, lane, wave);
        const real vn = w6[0];
        const real vtol = 1e-4 * vn;
        if (vn > 1e-12 * fmax(1.0, qnorm) && w6[1] <= vtol) {   // block-uniform
          real ssum = 0, bad = 0;
#pragma unroll 1
          for (int q = wave; q < n_site; q += NWV) {   // site rows
            RELANE();
            const int m = __builtin_amdgcn_readfirstlane(q / nct), c = q - m * nct;
            const int tt = 16 * c + t;
            real pk = M::big;
            if (A.peak && tt < Tm) { const double pv = A.peak[(size_t)b * Tm + tt]; pk = pv < M::big ? pv * A.peak_scale : M::big; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned i = sidx2(m, c, r);
              const int j = 16 * m + M::rowof(g, r);
              const real v2_ = at(Y2, i) - (real)at(Y2P, i);
              const int ty = RowTy[j];
              if (ty == kRowBox) { ssum += RowLim[j] * fmax(v2_, 0.0); if (v2_ < -vtol) bad = 1; }
              else if (ty == kRowPeak) {
                if (pk < M::big) ssum += pk * fmax(v2_, 0.0); else if (v2_ > vtol) bad = 1;
                if (v2_ < -vtol) bad = 1;
              } else if (ty == kRowSocRe) {
                const unsigned i2 = sidx2(m, c, (r + 1) & 3);
                const real vi = at(Y2, i2) - (real)at(Y2P, i2);
                ssum += RowLim[j] * sqrt(v2_ * v2_ + vi * vi);
              } else if (ty == kRowSocIm) {
              } else if (fabs(v2_) > vtol) bad = 1;   // free / quadratic rows admit no ray
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
              if (c < nct) {   // the snapshot only covers the columns that hold periods
                const unsigned i = fidx(e, c, r);
                vv[c] = at(Y1s, i) - (real)at(Y1P, i); lbv[c] = at(LBs, i); ubv[c] = at(UBs, i);
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
        for (int q = wave; q < n_tile; q += NWV) {
          RELANE();
          if (q < n_site) {
            const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned i = sidx2(mo, c, r); at(Y2P, i) = (float)at(Y2, i); }
          } else {
            const int qe = q - n_site;
            const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned i = fidx(e, c, r); at(Y1P, i) = (float)at(Y1s, i); }
          }
        }
        have_prev = true;
      }
      const real score = fmax(pri / fmax(eps_p, 1e-300), dua / fmax(eps_d, 1e-300));
      if (score < kStallGain * best_score) { best_score = score; best_it = it; }
      const bool inacc = inaccurate_ok<real>(pri, dua, npri, ndua, A.eps_abs, A.eps_rel, A.inacc_floor);
      const bool stalled = A.stall_iters > 0 && it - best_it >= A.stall_iters && score <= kStallNear * best_score;   // acn_qp_tiled.hpp
      bool hand_over = false;
      if (!done && pass == 0 && A.polish_iters > 0 && it >= A.polish_iters) {   // block-uniform
        // rows the polish's Schur system would have (acn_qp_tiled.hpp): more than it holds -> the ADMM goes on
        real cnt = 0;
        const real ytol = 1e-9 * fmax(1.0, qnorm);
#pragma unroll 1
        for (int q = wave; q < n_site; q += NWV) {
          RELANE();
          const int m = __builtin_amdgcn_readfirstlane(q / nct), c = q - m * nct;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * m + M::rowof(g, r);
            const int ty = RowTy[j];
            const real yr = at(Y2, sidx2(m, c, r)), yi = at(Y2, sidx2(m, c, (r + 1) & 3));
            const bool disc = ty == kRowSocRe;
            const real mag = disc ? sqrt(yr * yr + yi * yi) : yr;
            const bool counts = (disc || ty == kRowBox || ty == kRowPeak) && 16 * c + t < Tm && mag > ytol;
            cnt += counts ? (disc ? 2.0 : 1.0) : 0.0;
          }
        }
        cnt = wave_sum<real>(cnt);
        __syncthreads();
        if (lane == 0) SC[wave] = cnt;
        __syncthreads();
        real tot = 0;
        for (int wv = 0; wv < NWV; ++wv) tot += SC[wv];
        __syncthreads();
        hand_over = tot + 8.0 <= (real)A.pol_rows;
      }
      if (done) {
      } else if (hand_over) {
        status = kStatusPolish;   // not converged after polish_iters iterations: the polish kernel takes over (acn_qp_polish.hpp)
        done = true;
      } else if (it >= max_iter_p || stalled) {
        done = true;
        if (inacc) status = 5;
      } else if (adapt_p > 0 && it % adapt_p == 0) {
        const real sp = pri / fmax(npri, 1e-12), sd = dua / fmax(ndua, 1e-12);
        const real ratio = sqrt(sp / fmax(sd, 1e-30));
        const real tol_eff = A.adapt_tol * (1.0 + (real)n_adapt * (1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = fmin(fmax(rho * ratio, 1e-6), 1e6);
          set_dj();
          // r0 and u depend on rho; the fixed-point map changed: restart the ring from the current (z, y)
#pragma unroll 1
          for (int q = wave; q < n_tile; q += NWV) {
            RELANE();
            if (q < n_site) { const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct; if (aa_m > 0) reset_site(mo, c); }
            else { const int qe = q - n_site; const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct; reset_evse(e, c); }
          }
          if (aa_m > 0) {
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < AMX * AMX + AMX; k += 64) AaH[k] = 0;
          }
          __syncthreads();
        }
      }
    }
    STAMP(8);   // residual check (amortised)
  }
  // ---- results of this pass: the feasible iterate z1 is the schedule (kept if it beats the earlier passes) ------
  it_total += it;
  __syncthreads();
  if (pass == 0 || status_rank(status) > status_rank(best_status)) {   // block-uniform
  best_status = status;
  real ol = 0;
#pragma unroll 1
  for (int q = wave; q < n_tile; q += NWV) {
    RELANE();
    if (q < n_site) {
      if (A.y_out && (!A.y_for_polish_only || status == kStatusPolish)) {   // site-row multipliers in the caller's row order and units
        const int mo = __builtin_amdgcn_readfirstlane(q / nct), c = q - mo * nct;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * mo + M::rowof(g, r), tt = 16 * c + t;
          const int ja = A.rowabi[j];
          if (ja >= 0 && tt < Tm)
            A.y_out[((size_t)b * A.Mg + ja) * Tm + tt] = at(Y2, sidx2(mo, c, r)) * static_cast<const real*>(A.rowscale)[j];
        }
      }
    } else {
      const int qe = q - n_site;
      const int e = __builtin_amdgcn_readfirstlane(qe / nct), c = qe - e * nct;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ev = 16 * e + M::rowof(g, r), tt = 16 * c + t;
        if (ev < N && tt < Tm) {
          const unsigned i = fidx(e, c, r);
          const real z = at(Z1s, i);
          A.x[((size_t)b * N + ev) * Tm + tt] = z;
          ol += (0.5 * pd_user * z + at(Qs, i)) * z;
        }
      }
    }
  }
  ol = wave_sum<real>(ol);
  if (lane == 0) SC[wave] = ol;
  __syncthreads();
  if (tid == 0) {
    real o = 0;
    for (int wv = 0; wv < NWV; ++wv) o += SC[wv];
    A.status[b] = status; A.pri[b] = pri; A.dua[b] = dua; A.obj[b] = o;
    if (status == kStatusPolish) A.pol_list[atomicAdd(A.pol_count, 1)] = b;
  }
  }
  if (tid == 0) A.iters[b] = it_total;
#ifdef ACNQP_STAMPS
  if (lane == 0 && b < 1024 && wave < 16)
    for (int k = 0; k < 12; ++k) g_stamps[(b * 16 + wave) * 12 + k] = st_acc[k];
#endif
  if (!retry_wanted(pass, A.retry_passes, status, it, A.stall_iters, A.adapt_every)) break;
  __syncthreads();
  }   // passes
  }   // work queue
}

#undef RELANE
}  // namespace acnqp
