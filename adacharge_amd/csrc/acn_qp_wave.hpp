// Wave-per-problem batched MPC QP solver for gfx950: ONE wavefront solves one problem, lane = EVSE, the whole horizon
// of an EVSE (<= 12 periods) in that lane's registers; four independent waves per workgroup (one per SIMD, 512
// registers each) share the site's MFMA fragments in LDS.  Same algorithm, parameters and exits as the register-
// resident kernel of acn_qp_tiled.hpp (twins: oracle/admm_port.c, oracle/admm_ref.py) -- a different data layout:
//
//   * the session energy rows (water-filling, the dependent chain that was a third of the tiled kernel's iteration)
//     are LANE-LOCAL: twelve registers, no DPP / permlane reduction, one pass serves the wave's 64 sessions;
//   * nothing is done redundantly: the site-row work (w^, e^, h^, the projection onto the discs) once per problem, not
//     once per wave of the problem; no workgroup barrier anywhere in the solver loop (the waves never meet);
//   * the cross-EVSE products stay on the matrix pipe: the iterate goes EVSE layout -> MFMA operand layout and back
//     through a wave-private LDS scratch (24 ds_write_b64 + 28 ds_read_b64 per iteration),
//       P    (rows x periods)  = Ghat r0            16 x v_mfma_f64_16x16x4  (B operand = r0 through the scratch)
//       w^   = Q'(rho z2 - y2),  G x~ = Q h^         4 + 4                    (accumulator tiles as B operands)
//       corr (periods x EVSEs) = e^' Ghat            16                       (A operand = the e^ accumulator tile)
//     with the A / B fragments of Ghat and Q read from LDS (20 KB per workgroup, loaded once per launch).
//
// Several waves per problem (NPW = 2 or 4): wave h of the group holds periods TSV h ... TSV h + TSV - 1 of every EVSE.
// The products and the site rows are per period, so they stay wave-local (each wave its own transposes, MFMA chains and
// site-row tile); what couples the parts are sums over a session's window (the water-filling: two per-lane values per
// pass), the Anderson event's seven dot products, the residual check's maxima and the queue position -- exchanged through
// a 2 KB LDS mailbox per wave with a sequence flag (grp_xchg: a wave's in-order LDS writes, then its flag; no workgroup
// barrier: another group of the workgroup is never involved).  Every wave reads all the group's mailboxes in part order,
// so sums are the same bits everywhere and every wave takes the same branches.
//
// Instantiations <AM, NPW, TSV, MT> (acn_qp_wave.hip routes by shape; N <= 64 EVSEs, one session slot per EVSE, box /
// disc / peak rows, the load-flattening and the demand-charge row):
//   <5, 1, 12, 1>  horizon <= 12, <= 16 site rows            one wave per problem, four problems per workgroup (the headline)
//   <5, 2, 12, 1>  horizon 13 ... 24, <= 16 site rows        two waves of twelve periods
//   <5, 2,  6, 2>  horizon <= 12, 17 ... 32 site rows        two waves of six periods, two row tiles (a lane's state halves:
//                                                            room for the second site-row tile without a spill)
//   <5, 4,  6, 2>  horizon 13 ... 24, 17 ... 32 site rows    four waves of six periods: one problem per workgroup
//   <5, 4, 12, 1>  horizon 33 ... 48, <= 16 site rows        four waves of twelve periods: one problem per workgroup
// Everything else stays with acn_qp_tiled.hpp / acn_qp_long.hpp (ACNQP_NO_WAVE=1, ACNQP_NO_WAVE2=1: the A/B switches of
// tests/test_wave_kernel.py).
#pragma once
#include "acn_qp_tiled.hpp"

namespace acnqp {

constexpr int kWaveTS = 12;   // period slots per lane
constexpr int kWaveXS = 17;   // row stride (doubles) of the transpose scratch: 64 rows (EVSEs) x 16 periods, odd stride
constexpr int kWaveNW = 4;    // waves (= problems in flight) per workgroup
constexpr int kWaveAM = 5;    // Anderson columns compiled in

// LDS carve-up in doubles.  Shared: the MFMA fragments and the per-row constants.  Per wave: the transpose scratch, the
// rho-dependent row factors, the Gram matrix of the Anderson ring, the certificate's dual snapshot (floats) and the dF
// ring (floats; the dG ring lives in registers).
struct WaveLds {
  int fragp, fragx, fragq, rowc, wave0, wstride;   // offsets in doubles
  int xt, rowd, aah, snap, hist, total;            // per-wave offsets (relative to the wave's region), total in doubles
  int xch;                                         // NPW >= 2: the wave's mailbox (two buffers of [2][64] doubles) + its flag
  int xs;                                          // row stride of the transpose scratch (doubles)
  __host__ __device__ WaveLds(int accel_mem, int npw = 1, int mt = 1, int tsv = kWaveTS) {
    int o = 0;
    fragp = o; o += mt * 16 * 64;         // Ghat as A operand of P = Ghat r0: [row tile][k-step][lane]
    fragx = o; o += 4 * 4 * mt * 64;      // Ghat as B operand of corr = e^' Ghat: [EVSE tile][k-step (4 per row tile)][lane]
    fragq = o; o += mt * mt * 2 * 4 * 64; // Q' and Q as A operands (the tiled kernel's fragQ: [mo][mi][2][4][64])
    rowc = o;  o += mt * (16 + 16 + 8);   // eigenvalues, limits, row types (ints, lane order)
    wave0 = o;
    int w = 0;
    xs = tsv > 8 ? kWaveXS : 9;           // 64 EVSE rows x 16 (8) period columns, odd stride
    xt = w;   w += 64 * xs;
    rowd = w; w += 16 * mt;               // rho / (a + rho lam)
    aah = w;  w += kWaveAM * kWaveAM + kWaveAM + 1;
    w = (w + 1) & ~1;
    snap = w; xch = w;
    w += npw >= 2 ? 2 * 2 * 64 + 2 : 64 * 16 / 2;   // one wave per problem: the certificate snapshot, floats [chunk of 4][lane][4];
                                                    // two: the mailbox (the snapshot lives in registers there)
    w = (w + 1) & ~1;
    hist = w; w += accel_mem * 64 * 16 / 2;
    wstride = (w + 1) & ~1;
    total = wave0 + kWaveNW * wstride;
  }
};

// one queue position per WAVE: lane 0 fetches, the wave reads it as a scalar (no barrier: the waves are independent)
__device__ inline int wave_queue_next(int32_t* queue, int B, int round, int wave, int lane) {
  if (queue == nullptr) {
    const int pos = (int)blockIdx.x * kWaveNW + wave;
    return round == 0 && pos < B ? pos : -1;
  }
  int v = 0;
  if (lane == 0) v = atomicAdd(queue, 1);
  const int pos = __builtin_amdgcn_readfirstlane(v);
  return pos < B ? pos : -1;
}

__device__ inline void wave_lds_sync() {   // this wave's LDS writes are visible to its other lanes (in-order LDS; compiler fence)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int AM, int NPW, int TSV, int MT, bool PROX>
__global__ __launch_bounds__(kWaveNW * 64, 1) void admm_wave_kernel(const TiledArgs A_kernarg) {
  static_assert(NPW == 1 || NPW == 2 || NPW == 4, "one, two or four waves per problem");
  static_assert((TSV == 12 && MT == 1) || (TSV == 6 && MT == 2 && NPW >= 2), "instantiated: 12 periods x one row tile, 6 periods x two row tiles");
  using M = Mfma<double>;
  using vec4 = typename M::vec4;
  typedef double real;
  constexpr int TS = TSV, XS = TSV > 8 ? kWaveXS : 9, XC = TSV > 8 ? 16 : 8;   // period slots, scratch stride, its live columns
  constexpr int SR = 4 * MT, KS = 4 * MT;                                         // site-row registers per lane, k-steps of corr
#define BIGC (scalar_const(1e300))
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* sm = reinterpret_cast<real*>(smem_raw);
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int g = lane >> 4, tc = lane & 15;   // MFMA operand coordinates of the lane
  const int half = wave & (NPW - 1);            // which part of the horizon this wave holds (scalar; 0 with one wave per problem)
  const int tb = TSV * half;                    // its first period
  int xseq = 0;                                 // mailbox sequence number (the partner counts the same exchanges)
  bool xbroken = false;                         // a wait ran into its bound (never expected): no further waits

  // ---- once per workgroup: the site's fragments and row constants -> LDS ------------------------------------------------
  {
    const WaveLds L0(0, NPW, MT, TSV);
    const int NP = A_kernarg.NP;
    const real* Gh = static_cast<const real*>(A_kernarg.Ghat);
    const real* FQg = static_cast<const real*>(A_kernarg.fragQ);
    for (int k = tid0; k < MT * 16 * 64; k += kWaveNW * 64) {
      const int m = k >> 10, s = (k >> 6) & 15, l = k & 63, gg = l >> 4, tt = l & 15;
      sm[L0.fragp + k] = Gh[(size_t)(16 * m + tt) * NP + 4 * s + gg];                    // A[i = row 16 m + tt][k = EVSE 4 s + gg]
    }
    for (int k = tid0; k < 4 * KS * 64; k += kWaveNW * 64) {
      const int ws = k >> 6, l = k & 63, gg = l >> 4, tt = l & 15;
      const int w = ws / KS, s = ws % KS;
      sm[L0.fragx + k] = Gh[(size_t)(16 * (s >> 2) + 4 * (s & 3) + gg) * NP + 16 * w + tt];   // B[k = row of k-step s][j = EVSE 16 w + tt]
    }
    for (int k = tid0; k < MT * MT * 2 * 4 * 64; k += kWaveNW * 64) sm[L0.fragq + k] = FQg[k];
    if (tid0 < 16 * MT) {
      sm[L0.rowc + tid0] = static_cast<const real*>(A_kernarg.lam)[tid0];
      sm[L0.rowc + 16 * MT + tid0] = static_cast<const real*>(A_kernarg.rowlim)[tid0];
      const int m_ = tid0 >> 4, g_ = (tid0 >> 2) & 3, r_ = tid0 & 3;
      reinterpret_cast<int*>(sm + L0.rowc + 32 * MT)[tid0] = A_kernarg.rowtype[16 * m_ + M::rowof(g_, r_)];
    }
    const WaveLds L1(min(A_kernarg.accel_mem, AM), NPW, MT, TSV);
    real* XT0 = sm + L1.wave0 + (size_t)wave * L1.wstride;
    for (int k = lane; k < 64 * XS; k += 64) XT0[k] = 0;   // the pad columns (periods 12 ... 15) stay zero for good
    if (NPW >= 2 && lane < 2) reinterpret_cast<int*>(XT0 + L1.xch + 2 * 2 * 64)[lane] = 0;   // the mailbox flag
  }
  __syncthreads();   // the only workgroup barrier of the kernel

  // ---- the group's mailboxes (NPW >= 2): my two per-lane values out, every part's in -------------------------------------
  // Protocol: write the values into buffer (seq & 1) of MY mailbox, then my flag = seq (LDS operations of a wave complete
  // in order); wait for the partner's flag >= seq; read ITS buffer.  Double buffering is enough: the partner raises its
  // flag to seq + 1 only after it has read my buffer of seq, and I write that buffer again at seq + 2.  Every wait is
  // bounded (a wave that never arrives would otherwise hang the queue for good).
  real* Xm = nullptr;
  const real* Xg = nullptr;   // mailbox of the group's first wave; wave k's is Xg + k * xstride
  int xstride = 0;
  if constexpr (NPW >= 2) {
    const WaveLds Lx(min(A_kernarg.accel_mem, AM), NPW, MT, TSV);
    Xm = sm + Lx.wave0 + (size_t)wave * Lx.wstride + Lx.xch;
    Xg = sm + Lx.wave0 + (size_t)(wave & ~(NPW - 1)) * Lx.wstride + Lx.xch;
    xstride = Lx.wstride;
  }
  // every part's two per-lane values, in part order (index `half` = my own): the same arrays in every wave of the group,
  // so that sums formed from them in index order are the same bits everywhere
  auto grp_xchg = [&](real va, real vb, real (&pa)[NPW], real (&pb)[NPW]) __attribute__((always_inline)) {
    if constexpr (NPW >= 2) {
      ++xseq;
      const int o = (xseq & 1) * 128;
      Xm[o + lane] = va;
      Xm[o + 64 + lane] = vb;
      // (LDS is one memory per CU and a wave's LDS operations complete in order: the compiler must keep the order -- the
      //  wavefront-scope fences -- and the hardware needs no wait between the values and the flag)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) *reinterpret_cast<volatile int*>(Xm + 256) = xseq;
#pragma unroll
      for (int k = 0; k < NPW; ++k) {   // every part's mailbox in part order, my own included (its flag is up already): no branch on `half`
        const real* Xk = Xg + (size_t)k * xstride;
        if (!xbroken) {
          int spins = 0;
          while (*reinterpret_cast<const volatile int*>(Xk + 256) - xseq < 0) {
            if (++spins > 64) __builtin_amdgcn_s_sleep(1);   // the partner is usually a few hundred cycles away: poll first
            if (spins > (1 << 22)) { xbroken = true; break; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pa[k] = Xk[o + lane];
        pb[k] = Xk[o + 64 + lane];
      }
    } else {
      pa[0] = va; pb[0] = vb;
    }
  };
  // per-lane sum / min-max over the problem's waves (identity with one wave)
  auto pl_sum2 = [&](real& va, real& vb) __attribute__((always_inline)) {
    if constexpr (NPW >= 2) {
      real pa[NPW], pb[NPW];
      grp_xchg(va, vb, pa, pb);
      va = pa[0]; vb = pb[0];
#pragma unroll
      for (int k = 1; k < NPW; ++k) { va += pa[k]; vb += pb[k]; }
    }
  };
  auto pl_minmax = [&](real& lo_, real& hi_) __attribute__((always_inline)) {
    if constexpr (NPW >= 2) {
      real pa[NPW], pb[NPW];
      grp_xchg(lo_, hi_, pa, pb);
#pragma unroll
      for (int k = 0; k < NPW; ++k) { lo_ = fmin(lo_, pa[k]); hi_ = fmax(hi_, pb[k]); }
    }
  };
  // up to 8 wave-uniform values at once: value j rides in lane j
  auto pu_sum = [&](real* d, int n) __attribute__((always_inline)) {
    if constexpr (NPW >= 2) {
      real v = 0, pa[NPW], pb[NPW];
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < n) v = lane == j ? d[j] : v;
      grp_xchg(v, 0.0, pa, pb);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < n) {
          real t_ = lane_value(pa[0], j);
#pragma unroll
          for (int k = 1; k < NPW; ++k) t_ += lane_value(pa[k], j);
          d[j] = t_;
        }
    }
  };
  auto pu_max = [&](real* d, int n) __attribute__((always_inline)) {
    if constexpr (NPW >= 2) {
      real v = 0, pa[NPW], pb[NPW];
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < n) v = lane == j ? d[j] : v;
      grp_xchg(v, 0.0, pa, pb);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < n) {
#pragma unroll
          for (int k = 0; k < NPW; ++k) d[j] = fmax(d[j], lane_value(pa[k], j));
        }
    }
  };

  for (int q_round = 0;; ++q_round) {   // work queue: this WAVE's (pair's) next problem
  int q_pos;
  if constexpr (NPW >= 2) {   // the group's first wave fetches; the position travels through the mailbox
    int mine = 0;
    if (half == 0) {
      if (A_kernarg.queue == nullptr) {
        const int pos = (int)blockIdx.x * (kWaveNW / NPW) + wave / NPW;
        mine = q_round == 0 && pos < queue_length(A_kernarg) ? pos : -1;
      } else {
        mine = wave_queue_next(A_kernarg.queue, queue_length(A_kernarg), q_round, wave, lane);
      }
    }
    real pa[NPW], pb[NPW];
    grp_xchg((real)mine, 0.0, pa, pb);
    q_pos = __builtin_amdgcn_readfirstlane((int)pa[0]);
  } else {
    q_pos = wave_queue_next(A_kernarg.queue, queue_length(A_kernarg), q_round, wave, lane);
  }
  if (q_pos < 0) break;
  int it_total = 0, best_status = 0;
  for (int pass = 0;; ++pass) {
  int b_ = q_pos;
  asm volatile("" : "+v"(b_));
  const int wg_ = __builtin_amdgcn_readfirstlane(b_);
  KernargPtr Ap = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(Ap));
  const auto& A = *Ap;
  const int b = __builtin_amdgcn_readfirstlane(A.order ? A.order[wg_] : wg_);
  if (A.resume) {   // the launch behind the polish kernel (wave-uniform)
    if (A.status[b] != kStatusPolish && pass == 0) break;
    if (pass == 0) it_total = A.iters[b];
  }
  const int max_iter_p = pass == 0 ? A.max_iter : min(A.max_iter, A.retry_max_iter);
  const int adapt_p = pass == 0 ? A.adapt_every : 0;
  const int N = A.N, Tm = A.Tm;
  const int aa_m = min(A.accel_mem, AM);
  const WaveLds L(aa_m, NPW, MT, TSV);
  const real* FragP = sm + L.fragp;
  const real* FragX = sm + L.fragx;
  const real* FQs = sm + L.fragq;
  const real* RowLam = sm + L.rowc;
  const real* RowLim = RowLam + 16 * MT;
  const int* RowTy = reinterpret_cast<const int*>(RowLam + 32 * MT);
  real* Wv = sm + L.wave0 + (size_t)wave * L.wstride;
  real* XT = Wv + L.xt;
  real* RowDj = Wv + L.rowd;
  real* AaH = Wv + L.aah;
  float* Snap = reinterpret_cast<float*>(Wv + L.snap);   // [chunk][lane][4]: chunks 0..2 = y1 of the lane's EVSE, chunk 3 = y2 (C layout)
  float* HistF = reinterpret_cast<float*>(Wv + L.hist);  // dF ring: [slot][chunk][lane][4]
  const real* Gm = static_cast<const real*>(A.G);
  auto row_types = [&](int m, int (&ty)[4]) __attribute__((always_inline)) {
    const int4 v = *reinterpret_cast<const int4*>(RowTy + (m * 4 + g) * 4);
    ty[0] = v.x; ty[1] = v.y; ty[2] = v.z; ty[3] = v.w;
  };

  // ---- layout changes through the wave's scratch -----------------------------------------------------------------------
  // EVSE layout (lane = EVSE, register = period) -> C-layout tile (rows x periods) of Amat v, Amat given as A fragments
  auto evse_to_rows = [&](const real (&v)[TS], const real* frag, vec4 (&out)[MT]) __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < TS; ++t) XT[lane * XS + t] = v[t];
    wave_lds_sync();
    // every operand requested before the first MFMA (no branch in the chain: EVSEs beyond N hold zeros); two
    // accumulators per row tile, so that a product does not wait for the previous one's result
    real bop[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if constexpr (XC == 16) bop[s] = XT[(4 * s + g) * XS + tc];
      else bop[s] = tc < XC ? XT[(4 * s + g) * XS + (tc < XC ? tc : 0)] : 0.0;   // (the scratch holds 8 period columns)
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      real aop[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) aop[s] = frag[(m * 16 + s) * 64 + lane];
      vec4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        acc0 = M::mma(aop[s], bop[s], acc0);
        acc1 = M::mma(aop[s + 1], bop[s + 1], acc1);
      }
      out[m] = acc0 + acc1;
    }
  };
  // C-layout tiles c4 (rows x periods) -> EVSE layout of Bmat' c4, Bmat (rows x EVSEs) given by bfrag(tile, k-step)
  auto rows_to_evse = [&](const vec4 (&c4)[MT], auto&& bfrag, real (&out)[TS]) __attribute__((always_inline)) {
    real bf[4][KS];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int s = 0; s < KS; ++s) bf[w][s] = bfrag(w, s);
    vec4 acc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[w] = vec4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[w] = M::mma(c4[s >> 2][s & 3], bf[w][s], acc[w]);   // (four independent chains, interleaved)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int r = 0; r < 3; ++r) {   // (period g + 4 r of EVSE 16 w + tc)
        if constexpr (TS >= 12) XT[(16 * w + tc) * XS + g + 4 * r] = acc[w][r];
        else if (4 * r < TS) { if (g + 4 * r < TS) XT[(16 * w + tc) * XS + g + 4 * r] = acc[w][r]; }
      }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < TS; ++t) out[t] = XT[lane * XS + t];
  };
  auto frag_ghat = [&](int w, int s) __attribute__((always_inline)) -> real { return FragX[(w * KS + s) * 64 + lane]; };
  auto frag_g = [&](int w, int s) __attribute__((always_inline)) -> real { return Gm[(size_t)(16 * (s >> 2) + 4 * (s & 3) + g) * A.NP + 16 * w + tc]; };

  // ---- problem data -> registers (EVSE layout) ------------------------------------------------------------------------
  const bool act = lane < N;
  real x[TS], z1[TS], y1[TS], qv[TS], lbv[TS], ubv[TS];
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    const bool ok = act && tb + t < Tm;
    const size_t idx = ((size_t)b * N + (ok ? lane : 0)) * Tm + (ok ? tb + t : 0);
    lbv[t] = ok ? A.lb[idx] : 0.0;
    ubv[t] = ok ? A.ub[idx] : 0.0;
    qv[t] = ok ? A.q[idx] : 0.0;
    if (ubv[t] < lbv[t]) ubv[t] = lbv[t];
    x[t] = 0; z1[t] = 0; y1[t] = 0;
  }
  const bool eq = A.s_eq[b] != 0;
  unsigned swm = 0;   // bit t: period t lies in the session's window
  int smode = 4;      // 0: root-find each iteration, 2: pinned at ub, 3: pinned at lb, 4: no session
  real scap = 0, mu = 0;
  unsigned imask = 0;   // bit t: period t was strictly inside its bounds at the end of the previous projection (project_B)
  bool empty_set = false;
  {
    const size_t sidx = (size_t)b * N + (act ? lane : 0);   // (K == 1)
    const int off = act ? A.s_off[sidx] : 0;
    const int len = act ? A.s_len[sidx] : 0;
    real sl = 0, su = 0;
#pragma unroll
    for (int t = 0; t < TS; ++t)
      if (tb + t >= off && tb + t < off + len && tb + t < Tm) { swm |= 1u << t; sl += lbv[t]; su += ubv[t]; }
    pl_sum2(sl, su);   // (the window may span both halves)
    scap = act ? A.s_cap[sidx] : 0.0;
    if (len > 0) {
      smode = 0;
      const real slack = 64.0 * M::proj_tol * fmax(1.0, fabs(scap));
      if (sl > scap + slack) empty_set = true;
      if (eq && su < scap - slack) empty_set = true;
      if (eq && scap >= su) smode = 2;
      else if (scap <= sl) smode = 3;
    }
  }
  // the peak limit of the lane's period column (C layout), scaled like its row
  real pk_lane = BIGC;
  if (A.peak && tc < TS && tb + tc < Tm) {
    const double pv = A.peak[(size_t)b * Tm + tb + tc];
    pk_lane = pv < 1e300 ? pv * A.peak_scale : BIGC;
  }

  const real pd_user = uniform_scalar(A.pdiag[b]);
  // load flattening's aggregate-power row lives in equilibrated units z' = s z: 1/2 lf z^2 = 1/2 (lf / s^2) z'^2
  // (PROX = false: sites without a prox row -- the headline -- carry none of this code: its mere presence cost 4 %)
  const real lfb = PROX ? uniform_scalar(A.lf ? A.lf[b] / (A.flat_scale * A.flat_scale) : 0.0) : 0.0;
  // ... the demand-charge row's too: dc max(z) = (dc / s) max(z')
  const real dcb = PROX ? uniform_scalar(A.dc ? A.dc[b] / A.max_scale : 0.0) : 0.0;
  const real dfl = PROX ? uniform_scalar(A.dfloor ? A.dfloor[b] * A.max_scale : 0.0) : 0.0;
  real tau_max = 0;   // warm start of the demand-charge level
  const real sigma = A.sigma, alpha = A.alpha;
  real rho = A.rho0;
  if (pass > 0) {
    rho = A.retry_rho;
    for (int k = 1; k < pass; ++k) rho *= 4.0;
  }
  real qnorm, pd;
  bool plain_windows;
  {
    real f1 = 0, f2 = 0, f3 = 0;
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      f1 = fmax(f1, fabs(qv[t]));
      f2 = fmax(f2, ubv[t]);
      if (!((swm >> t) & 1u)) f3 = fmax(f3, fmax(fabs(lbv[t]), fabs(ubv[t])));
    }
    f1 = wave_max<real>(f1); f2 = wave_max<real>(f2); f3 = wave_max<real>(f3);
    { real f[3] = {f1, f2, f3}; pu_max(f, 3); f1 = f[0]; f2 = f[1]; f3 = f[2]; }
    plain_windows = uniform_scalar(f3) == 0.0;
    qnorm = uniform_scalar(f1);
    pd = uniform_scalar(effective_pdiag<real>(pd_user, A.reg_rel, qnorm, uniform_scalar(f2), A.horizon[b], PROX && ((lfb > 0.0) | (dcb > 0.0))));
    if (__any(empty_set)) {   // a session cannot meet its energy row inside its own bounds (wave-uniform)
#pragma unroll
      for (int t = 0; t < TS; ++t)
        if (act && tb + t < Tm) A.x[((size_t)b * N + lane) * Tm + tb + t] = 0;
      if (A.y_out && !A.y_for_polish_only && half == 0)
        for (int k = lane; k < A.Mg * Tm; k += 64) A.y_out[(size_t)b * A.Mg * Tm + k] = 0;
      if (lane == 0 && half == 0) {
        A.status[b] = 4; A.iters[b] = 0;
        A.pri[b] = 1e300; A.dua[b] = 1e300; A.obj[b] = 0;
      }
      break;
    }
  }

  // site-row state (C layout: lane (g, tc), register r <-> row g + 4 r, period tc)
  real z2[SR], y2[SR], gx[SR];   // (register 4 m + r <-> row 16 m + g + 4 r)
#pragma unroll
  for (int r = 0; r < SR; ++r) { z2[r] = 0; y2[r] = 0; gx[r] = 0; }

  rho = uniform_scalar(rho);
  real a = sigma + pd + rho, inv_a = uniform_scalar(1.0 / a), inv_rho = uniform_scalar(1.0 / rho);
  __builtin_amdgcn_wave_barrier();
  if (lane < 16 * MT) RowDj[lane] = rho / (a + rho * RowLam[lane]);
  wave_lds_sync();

  int status = 2, it = 0, n_adapt = 0, best_it = 0;
  // (counters, not `it % every`: a modulo by a kernel argument was a scalar load and an integer division per iteration)
  const int check_every = A.check_every;
  int next_check = check_every, next_aa = kAaPeriod;
  real best_score = BIGC;
  real pri = BIGC, dua = BIGC;
  bool done = false, have_prev = false;
  float sn1[TS], sn2[SR];   // NPW >= 2: the certificate's dual snapshot (one wave per problem keeps it in LDS: Snap)
#pragma unroll
  for (int t = 0; t < TS; ++t) sn1[t] = 0.f;
#pragma unroll
  for (int r = 0; r < SR; ++r) sn2[r] = 0.f;

  // ---- Anderson acceleration state (wave-uniform scalars; vectors: EVSE layout [0, TS) then the site tile [TS, TS + 4)) ----
  constexpr int DV = TS + SR, DVP = 16;   // the Anderson state per lane; its ring row padded to four float4 chunks
  static_assert(DV <= DVP, "ring row");
  real up[DV];
  float fp[DV], cp[DV];
  float hg[AM][DV];   // dG ring (registers)
#pragma unroll
  for (int k = 0; k < DV; ++k) {
    up[k] = 0; fp[k] = 0.f; cp[k] = 0.f;
#pragma unroll
    for (int j = 0; j < AM; ++j) hg[j][k] = 0.f;
  }
  int aa_cnt = 0, aa_head = 0, aa_cool = 0, aa_pen = 1;
  unsigned aa_valid = 0;
  bool aa_have_prev = false, aa_was = false;
  real fn_prev = 0;
  if (aa_m > 0) {
    for (int k = lane; k < AM * AM + AM; k += 64) AaH[k] = 0;
    for (int k = lane; k < aa_m * 64 * 16; k += 64) HistF[k] = 0.f;   // dead ring slots are read (zero coefficient): finite numbers
    wave_lds_sync();
  }

#ifdef ACNQP_STAMPS
  unsigned long long st_acc[24] = {0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  // ---- projection onto B = bounds /\ the session's energy row: exact water-filling, lane-local -------------------------
  // z = clip(zh - m) with g(m) = sum_t clip(zh_t - m) = cap: safeguarded Newton on the piecewise-linear g, warm-started
  // at the previous iteration's m.  `between` is independent work (the site-row projection) placed in front of it.
  auto project_B = [&](const real (&zin)[TS], auto&& between) __attribute__((always_inline)) {
    between();
    STAMP(5);   // site-row projection
    const real cap = scap;
    const real tol = M::proj_tol * fmax(1.0, fabs(cap));
    bool need = smode == 0;
    // one clip serves every mode: the multiplier being searched, -big (pinned at ub), +big (pinned at lb), 0 (no session)
    real m = smode == 0 ? mu : 0.0;
    m = smode == 2 ? -BIGC : m;
    m = smode == 3 ? BIGC : m;
    // Start of the search: the multiplier that is EXACT if the session's active set is the one the previous projection
    // ended with (imask: bit t = period t was strictly inside its bounds; z1 still holds that projection, so a clipped
    // period contributes its bound): m = (sum_t [inside ? zin_t : z1_t] - cap) / #inside.  The first evaluation below then
    // usually only verifies it (passes per iteration 1.94 -> ~1.1); where the set did change it is a start like any other.
    if (plain_windows) {   // (wave-uniform; with general windows the sum would need the window mask: the plain start)
      real sp = 0;
#pragma unroll
      for (int t = 0; t < TS; ++t) sp += ((imask >> t) & 1u) ? zin[t] : z1[t];
      int ni = __builtin_popcount(imask);
      if constexpr (NPW >= 2) { real nr = (real)ni; pl_sum2(sp, nr); ni = (int)nr; }
      real mp = (sp - cap) * rcp_small((float)(ni > 0 ? ni : 1));
      mp = (!eq & (mp < 0.0)) ? 0.0 : mp;
      m = (need & (ni > 0)) ? mp : m;
    }
    real lo = eq ? -BIGC : -1.0;   // inequality: m >= 0, so (-1, .) brackets m = 0
    real hi = BIGC;
    int guard = 0;
    // Every pass EVALUATES g at the current m and leaves z1 = clip(zin - m): the pass that finds |g(m) - cap| <= tol for
    // every session has already written the projection.  (The tiled kernel avoids the verifying evaluation with a
    // "no period changed its piece" test, because an evaluation there costs lane exchanges; here it is twelve lane-local
    // clips, cheaper than the test.)
    for (;;) {
      ++guard;
#ifdef ACNQP_STAMPS
      st_acc[8] += 1000;   // diagnostic build: slot 8 counts water-filling passes (x1000)
#endif
      real gl = 0;
      unsigned im = 0;
      if (plain_windows) {   // outside the window lb = ub = 0: no mask needed (same bits as the masked form)
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const real v = zin[t] - m;
          const real z = fmin(fmax(v, lbv[t]), ubv[t]);
          z1[t] = z;
          gl += z;
          im |= ((v > lbv[t]) & (v < ubv[t])) ? (1u << t) : 0u;
        }
      } else {
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const bool inw = (swm >> t) & 1u;
          const real v = zin[t] - (inw ? m : 0.0);
          const real z = fmin(fmax(v, lbv[t]), ubv[t]);
          z1[t] = z;
          gl += inw ? z : 0.0;
          im |= (inw & (v > lbv[t]) & (v < ubv[t])) ? (1u << t) : 0u;
        }
      }
      imask = im;
      float nl = (float)__builtin_popcount(im);
      if constexpr (NPW >= 2) { real nr = (real)nl; pl_sum2(gl, nr); nl = (float)nr; }
      const real d = gl - cap;
      const real big_ = BIGC;
      const bool fin = (fabs(d) <= tol) | (!eq & (m <= 0.0) & (d <= 0.0)) | (guard > ACNQP_GUARD_MAX);
      need = need & !fin;
      if (!__any(need)) break;
      const bool dpos = d > 0;
      lo = (need & dpos) ? m : lo;
      hi = (need & !dpos) ? m : hi;
      const bool open = need & (nl <= 0.f) & !((lo > -big_) & (hi < big_));
      if (__any(open)) {   // flat piece with an open bracket (rare): the true bracket ends, so that the fallback bisects
        real lo_l = big_, hi_l = -big_;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const bool inw = (swm >> t) & 1u;
          lo_l = inw ? fmin(lo_l, zin[t] - ubv[t]) : lo_l;
          hi_l = inw ? fmax(hi_l, zin[t] - lbv[t]) : hi_l;
        }
        pl_minmax(lo_l, hi_l);
        lo = open ? fmax(lo, lo_l) : lo;
        hi = open ? fmin(hi, hi_l) : hi;
      }
      const bool bracketed = (lo > -big_) & (hi < big_);
      const real mid = 0.5 * (lo + hi);
      const bool newton = nl > 0.f;
      real rc = rcp_small(newton ? nl : 1.f);
      asm volatile("" : "+v"(rc));
      real c_newton = m + d * rc, c_clamped = fmin(fmax(m + d, lo), hi);
      asm volatile("" : "+v"(c_newton), "+v"(c_clamped));
      real cand = newton ? c_newton : (bracketed ? mid : c_clamped);
      cand = (!eq & (cand < 0.0)) ? 0.0 : cand;   // inequality: multiplier >= 0
      real a_clamped = fmin(fmax(cand, lo), hi);
      asm volatile("" : "+v"(a_clamped));
      real alt = bracketed ? mid : a_clamped;
      alt = (!eq & (alt < 0.0)) ? 0.0 : alt;
      const bool inside = (cand > lo) & (cand < hi);
      cand = inside ? cand : alt;
      need = need & (cand != m);   // no representable progress: stop (z1 holds the clip at m)
      m = need ? cand : m;
    }
    mu = smode == 0 ? m : mu;
  };

  // ---- start (cold: the schedule that ignores the site rows; warm: A.warm_x / A.warm_y), as acn_qp_tiled.hpp ----------
  {
    const bool warm = pass == 0 && A.warm_x != nullptr && A.warm_y != nullptr;
    real zs[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      zs[t] = -scalar_const(kStartGain) * qv[t];
      if (warm) {
        const bool ok = act && tb + t < Tm;
        zs[t] = ok ? A.warm_x[((size_t)b * N + (ok ? lane : 0)) * Tm + (ok ? tb + t : 0)] : 0.0;
      }
    }
    project_B(zs, []() {});
    mu = 0;   // the multiplier of this one-off projection is no warm start
    real gty[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) gty[t] = 0;
    if (warm) {
      const real* RS = static_cast<const real*>(A.rowscale);
      vec4 yv[MT];
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        const int j = 16 * (r >> 2) + M::rowof(g, r & 3);
        const int ja = A.rowabi[j];
        const bool ok = ja >= 0 && tc < TS && tb + tc < Tm;
        y2[r] = ok ? A.warm_y[((size_t)b * A.Mg + (ok ? ja : 0)) * Tm + (ok ? tb + tc : 0)] / RS[j] : 0.0;
        yv[r >> 2][r & 3] = y2[r];
      }
      rows_to_evse(yv, frag_g, gty);
    }
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      x[t] = z1[t];
      y1[t] = -(qv[t] + pd * z1[t] + gty[t]);
      up[t] = z1[t] + y1[t] * inv_rho;
      cp[t] = 0.f;
    }
    vec4 g0[MT];
    evse_to_rows(z1, FragP, g0);
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      vec4 zt = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) zt = M::mma(FQs[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], g0[mi][s], zt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * mo + r;
        z2[k] = zt[r]; gx[k] = zt[r];
        if (!warm) y2[k] = 0;
        up[TS + k] = zt[r] + y2[k] * inv_rho;
        cp[TS + k] = 0.f;
      }
    }
  }

#ifdef ACNQP_STAMPS
  for (int k = 0; k < 24; ++k) st_acc[k] = 0;
  unsigned long long st_rt0, st_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
  st_t0 = st_prev;
#endif
  while (!done) {
    ++it;
    asm volatile("" : "+v"(swm));
    asm volatile("" : "+v"(smode));
    // ---- r0 = sigma x - q + rho z1 - y1;  P = Ghat r0;  w^ = Q'(rho z2 - y2) ----------------------------------------------
    real r0[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) r0[t] = sigma * x[t] - qv[t] + rho * z1[t] - y1[t];
    vec4 wh[MT];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      vec4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(FQs[(((mo * MT + mi) * 2 + 0) * 4 + s) * 64 + lane], rho * z2[4 * mi + s] - y2[4 * mi + s], acc);
      wh[mo] = acc;
    }
    STAMP(0);   // r0, w^ (4 MFMA)
    vec4 g0[MT];
    evse_to_rows(r0, FragP, g0);
    STAMP(1);   // EVSE -> rows: 12 writes, 32 reads, 16 MFMA
    // ---- e^ = w^ - D (g0 + Lam w^);  h^ = (g0 + Lam e^)/a -----------------------------------------------------------------
    vec4 eh[MT], hh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real w_ = wh[m][r];
        const real lam_ = RowLam[16 * m + M::rowof(g, r)];
        const real e_ = w_ - RowDj[16 * m + M::rowof(g, r)] * (g0[m][r] + lam_ * w_);
        eh[m][r] = e_;
        hh[m][r] = (g0[m][r] + lam_ * e_) * inv_a;
      }
    // ---- G x~ = Q h^ and the pre-projection point of the site rows (issued first: its VALU work overlaps the x~ MFMAs) ----
    real zhr[SR];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      vec4 zt = {0, 0, 0, 0};
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s) zt = M::mma(FQs[(((mo * MT + mi) * 2 + 1) * 4 + s) * 64 + lane], hh[mi][s], zt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * mo + r;
        gx[k] = alpha * zt[r] + (1.0 - alpha) * gx[k];
        zhr[k] = alpha * zt[r] + (1.0 - alpha) * z2[k] + y2[k] * inv_rho;
      }
    }
    STAMP(2);   // e^, h^, Q h^ (4 MFMA), zhr
    // ---- x~ = (r0 + Ghat' e^)/a, relaxation -------------------------------------------------------------------------------
    real zh[TS];
    {
      real corr[TS];
      rows_to_evse(eh, frag_ghat, corr);
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const real xn = (r0[t] + corr[t]) * inv_a;
        zh[t] = alpha * xn + (1.0 - alpha) * z1[t] + y1[t] * inv_rho;
        x[t] = alpha * xn + (1.0 - alpha) * x[t];
      }
    }
    STAMP(3);   // rows -> EVSE: 16 MFMA, 12 writes, 12 reads; relaxation
    const bool at_check = it == next_check;
    next_check += at_check ? check_every : 0;
    const bool check = at_check || it >= max_iter_p;

    // ---- Anderson acceleration event (acn_qp_tiled.hpp; state u = (zh, zhr)) ----------------------------------------------
    if constexpr (AM > 0) {
      const bool at_aa = it == next_aa;
      next_aa += at_aa ? kAaPeriod : 0;
      if (aa_m > 0 && at_aa) {
        const bool col = aa_have_prev;
        const int slot = aa_head;
        real f[DV];
        float cq[DVP];
#pragma unroll
        for (int k = DV; k < DVP; ++k) cq[k] = 0.f;
        real d[AM + 2];
        {
          real fa = 0;
#pragma unroll
          for (int k = 0; k < DV; ++k) {
            const real uk = k < TS ? zh[k < TS ? k : 0] : zhr[k < TS ? 0 : k - TS];
            f[k] = uk - up[k];
            fa += f[k] * f[k];
            cq[k] = (float)(f[k] - (real)fp[k]);
          }
          d[AM + 1] = fa;
        }
        {   // store the column pair in slot `slot` (speculatively: it only counts once marked live)
          float* hf = HistF + (size_t)slot * (4 * 64 * 4) + lane * 4;
#pragma unroll
          for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(hf + c * 256) = make_float4(cq[4 * c], cq[4 * c + 1], cq[4 * c + 2], cq[4 * c + 3]);
#pragma unroll
          for (int k = 0; k < DV; ++k) {
            const real uk = k < TS ? zh[k < TS ? k : 0] : zhr[k < TS ? 0 : k - TS];
            const float gv = (float)(uk - (up[k] + (real)cp[k]));
#pragma unroll
            for (int j = 0; j < AM; ++j) hg[j][k] = j == slot ? gv : hg[j][k];
          }
        }
        STAMP(12);   // event: f, column, ring stores
#pragma unroll
        for (int j = 0; j < AM; ++j) {   // dF_slot . dF_j for every ring slot j (own lane's entries: no sync needed)
          const int jj = j < aa_m ? j : aa_m - 1;
          const float* hj = HistF + (size_t)jj * (4 * 64 * 4) + lane * 4;
          real a1 = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 h4 = *reinterpret_cast<const float4*>(hj + c * 256);
            a1 += (real)cq[4 * c] * (real)h4.x + (real)cq[4 * c + 1] * (real)h4.y + (real)cq[4 * c + 2] * (real)h4.z + (real)cq[4 * c + 3] * (real)h4.w;
          }
          d[j] = a1;
        }
        {
          real a1 = 0;
#pragma unroll
          for (int k = 0; k < DV; ++k) a1 += (real)cq[k] * f[k];
          d[AM] = a1;
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) fp[k] = (float)f[k];
#pragma unroll
        for (int j = 0; j < AM + 2; ++j) d[j] = wave_sum<real>(d[j]);
        pu_sum(d, AM + 2);   // (two waves per problem: the partner's share)
        STAMP(13);   // event: dot products, wave sums
        const real fn = sqrt(d[AM + 1]);
        bool keep = col;
        if (aa_was && fn > scalar_const(kAaSafe) * fn_prev) {
          aa_cnt = 0; aa_head = 0; aa_valid = 0; keep = false;
          __builtin_amdgcn_wave_barrier();
          for (int k = lane; k < AM * AM + AM; k += 64) AaH[k] = 0;
          aa_cool = aa_pen;
          aa_pen = aa_pen < 64 ? 2 * aa_pen : 64;
        } else if (aa_cool > 0) --aa_cool;
        if (keep) {
          aa_valid |= 1u << slot;
          if (lane == 0) {
#pragma unroll
            for (int j = 0; j < AM; ++j) {
              if (!((aa_valid >> j) & 1u)) continue;
              AaH[slot * AM + j] = d[j];
              AaH[j * AM + slot] = d[j];
              if (j != slot) AaH[AM * AM + j] += d[j];
            }
            AaH[AM * AM + slot] = d[AM];
          }
          aa_head = slot + 1 == aa_m ? 0 : slot + 1;
          aa_cnt = aa_cnt < aa_m ? aa_cnt + 1 : aa_m;
        }
        wave_lds_sync();
        aa_have_prev = true; fn_prev = uniform_scalar(fn); aa_was = false;
        real dself = 0;
#pragma unroll
        for (int j = 0; j < AM; ++j) dself = j == slot ? d[j] : dself;
        if (aa_cnt > 0 && aa_cool == 0 && !check && dself > scalar_const(kAaDrift * kAaDrift) * d[AM + 1]) {
          // gamma = (H + eta I)^-1 b: Gauss-Jordan on the augmented AM x (AM + 1) system spread over the wave, lane 8 i + j
          // holding entry (i, j) (acn_qp_tiled.hpp; the whole system in every lane's registers measured no faster: 60
          // more live registers for 30 fewer lane exchanges)
          static_assert(AM <= 7, "one 8 x 8 lane tile holds the augmented system");
          const int gi = lane >> 3, gj = lane & 7;
          real tr = 0;
#pragma unroll
          for (int i = 0; i < AM; ++i) tr += AaH[i * AM + i];
          const real eta = scalar_const(kAaReg) * tr + scalar_const(1e-300);
          real ae = 0;
          if (gi < AM && gj <= AM) ae = gj < AM ? AaH[gi * AM + gj] : AaH[AM * AM + gi];
          if (gi < AM && gi == gj) ae = ((aa_valid >> gi) & 1u) ? ae + eta : 1.0;
#pragma unroll
          for (int k = 0; k < AM; ++k) {
            const real piv = lane_value(ae, 9 * k);
            const real rk = __shfl(ae, 8 * k + gj);
            const real ck = __shfl(ae, 8 * gi + k);
            const real rs = rk * rcp_nr(piv);
            ae = gi == k ? rs : ae - ck * rs;
          }
          real gam[AM];
#pragma unroll
          for (int j = 0; j < AM; ++j) gam[j] = __shfl(ae, 8 * j + AM);
          STAMP(14);   // event: bookkeeping, Gauss-Jordan
#pragma unroll
          for (int k = 0; k < DV; ++k) {
            real cor = 0;
#pragma unroll
            for (int j = 0; j < AM; ++j) cor += gam[j] * (real)hg[j][k];
            cp[k] = (float)cor;
            if (k < TS) zh[k < TS ? k : 0] -= (real)cp[k];
            else zhr[k < TS ? 0 : k - TS] -= (real)cp[k];
          }
          aa_was = true;
          STAMP(15);   // event: correction
        } else {
#pragma unroll
          for (int k = 0; k < DV; ++k) cp[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) up[k] = k < TS ? zh[k < TS ? k : 0] : zhr[k < TS ? 0 : k - TS];
      }
    }

    STAMP(4);   // Anderson event (amortised)
    project_B(zh, [&]() __attribute__((always_inline)) {
      // ---- site rows: projection of zhr onto C, y2 (branch-free over the row types, as acn_qp_tiled.hpp) ----------------
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int rty[4];
        row_types(m, rty);
        real scl[2], lim4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { lim4[r] = RowLim[16 * m + M::rowof(g, r)]; asm volatile("" : "+v"(lim4[r])); }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const real re = zhr[4 * m + 2 * pr], im = zhr[4 * m + 2 * pr + 1], lim = lim4[2 * pr];
          const real n2 = re * re + im * im;
          const bool clip = rty[2 * pr] == kRowSocRe && n2 > lim * lim;
          const real n2s = clip ? n2 : 1.0;
          const real f = lim * rsqrt_nr(n2s);
          scl[pr] = clip ? f : 1.0;
        }
        const real big_s = BIGC;
        const real quadf = rho / (rho + lfb);   // prox of 1/2 lf z^2 (the load-flattening row): z = zh rho / (rho + lf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real zh_ = zhr[4 * m + r];
          const int ty = rty[r];
          real fac = ((ty == kRowSocRe) | (ty == kRowSocIm)) ? scl[r >> 1] : 1.0;
          if constexpr (PROX) fac = ty == kRowQuad ? quadf : fac;
          real cap_ = ty == kRowBox ? lim4[r] : big_s;
          cap_ = ty == kRowPeak ? pk_lane : cap_;
          const real zn = fmin(zh_ * fac, cap_);
          y2[4 * m + r] = rho * (zh_ - zn);
          z2[4 * m + r] = zn;
        }
      }
      // ---- demand charge: prox of dc * max(max_t z_t, floor) over the whole horizon of the "max" row: z_t = min(zh_t, tau),
      // tau = max(floor, root of sum_t (zh_t - tau)+ = dc / rho) (Newton on a convex piecewise-linear function).  The row's
      // periods are the 16 lanes of a DPP row -- times the waves of the group: its sums cross the mailbox like a session's.
      if (PROX && A.dc != nullptr && dcb > 0.0) {   // wave-uniform
        real zv = 0;
        bool mine = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          int rty[4];
          row_types(m, rty);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (rty[r] == kRowMax) { mine = true; zv = z2[4 * m + r]; }   // = zh of that row (left unprojected above)
        }
        const bool live = (tc < TS) & (tb + tc < Tm);
        const real cw = dcb * inv_rho;
        real vmax = row_max<real>((mine & live) ? zv : -BIGC), vdummy = 0;
        pl_minmax(vdummy, vmax);
        real tau = tau_max;
        bool need = mine;
        int guard = 0;
        while (__any(need)) {
          ++guard;
          const real dd = zv - tau;
          const bool on = live & (dd > 0.0);
          real S = row_sum<real>(on ? dd : 0.0);
          real nr = (real)row_sum<float>(on ? 1.f : 0.f);
          pl_sum2(S, nr);
          const float nn = (float)nr;
          const real f = S - cw;
          const real tn = nn > 0.f ? tau + f * rcp_small(nn) : vmax - cw;
          const bool fin = (fabs(f) <= M::proj_tol * fmax(1.0, cw) * 16.0) | (tn == tau) | (guard > 60);
          tau = (need & !fin) ? tn : tau;
          need = need & !fin;
        }
        tau_max = tau;
        const real lev = fmax(tau, dfl);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          int rty[4];
          row_types(m, rty);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (rty[r] == kRowMax) {
              const real zh_ = z2[4 * m + r];
              const real zn = live ? fmin(zh_, lev) : zh_;
              y2[4 * m + r] = rho * (zh_ - zn);
              z2[4 * m + r] = zn;
            }
        }
      }
    });
#pragma unroll
    for (int t = 0; t < TS; ++t) y1[t] = rho * (zh[t] - z1[t]);

    STAMP(6);   // water-filling tail, z1, y1
    // ---- residuals, termination, rho adaptation (wave-uniform decisions) --------------------------------------------------
    if (check) {
      real v0 = 0, v1 = 0, v2 = 0, v4 = 0, v5 = 0;
      {
        real gty[TS];
        vec4 yv[MT];
#pragma unroll
        for (int r = 0; r < SR; ++r) yv[r >> 2][r & 3] = y2[r];
        rows_to_evse(yv, frag_g, gty);
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          v0 = fmax(v0, fabs(x[t] - z1[t]));
          v1 = fmax(v1, fabs(pd * x[t] + qv[t] + y1[t] + gty[t]));
          v2 = fmax(v2, fmax(fabs(x[t]), fabs(z1[t])));
          v4 = fmax(v4, fabs(pd * x[t]));
          v5 = fmax(v5, fabs(y1[t] + gty[t]));
        }
#pragma unroll
        for (int r = 0; r < SR; ++r) {
          v0 = fmax(v0, fabs(gx[r] - z2[r]));
          v2 = fmax(v2, fmax(fabs(gx[r]), fabs(z2[r])));
        }
      }
      real vm[4] = {wave_max<real>(v0), wave_max<real>(v1), wave_max<real>(v2), wave_max<real>(fmax(v4, v5))};
      pu_max(vm, 4);
      pri = uniform_scalar(vm[0]);
      dua = uniform_scalar(vm[1]);
      const real npri = uniform_scalar(vm[2]);
      const real ndua = fmax(uniform_scalar(vm[3]), qnorm);
      const real eps_p = A.eps_abs + A.eps_rel * npri;
      const real eps_d = A.eps_abs + A.eps_rel * ndua;
      if (pri <= eps_p && dua <= eps_d) { status = 1; done = true; }
      if (!done && have_prev) {
        // ---- primal infeasibility certificate (acn_qp_tiled.hpp / oracle/admm_port.c) ----------------------------------
        real w0 = 0, w1 = 0;
        real dv1[TS], dv2[SR];
        {
          vec4 dy[MT];
          if constexpr (NPW >= 2) {
#pragma unroll
            for (int r = 0; r < SR; ++r) dv2[r] = y2[r] - (real)sn2[r];
          } else {
            const float4 s2 = *reinterpret_cast<const float4*>(Snap + 3 * 256 + lane * 4);
            dv2[0] = y2[0] - (real)s2.x; dv2[1] = y2[1] - (real)s2.y; dv2[2] = y2[2] - (real)s2.z; dv2[3] = y2[3] - (real)s2.w;
          }
#pragma unroll
          for (int r = 0; r < SR; ++r) { dy[r >> 2][r & 3] = dv2[r]; w0 = fmax(w0, fabs(dv2[r])); }
          real gtv[TS];
          rows_to_evse(dy, frag_g, gtv);
          if constexpr (NPW >= 2) {
#pragma unroll
            for (int t = 0; t < TS; ++t) dv1[t] = y1[t] - (real)sn1[t];
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float4 s1 = *reinterpret_cast<const float4*>(Snap + c * 256 + lane * 4);
              dv1[4 * c] = y1[4 * c] - (real)s1.x; dv1[4 * c + 1] = y1[4 * c + 1] - (real)s1.y;
              dv1[4 * c + 2] = y1[4 * c + 2] - (real)s1.z; dv1[4 * c + 3] = y1[4 * c + 3] - (real)s1.w;
            }
          }
#pragma unroll
          for (int t = 0; t < TS; ++t) {
            w0 = fmax(w0, fabs(dv1[t]));
            w1 = fmax(w1, fabs(dv1[t] + gtv[t]));
          }
        }
        real wm[2] = {wave_max<real>(w0), wave_max<real>(w1)};
        pu_max(wm, 2);
        const real vn = uniform_scalar(wm[0]);
        const real atv = uniform_scalar(wm[1]);
        const real vtol = scalar_const(1e-4) * vn;
        if (vn > scalar_const(1e-12) * fmax(1.0, qnorm) && atv <= vtol) {
          real bad = 0, ssum = 0;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            int rty[4];
            row_types(m, rty);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const real v2_ = dv2[4 * m + r];
              const int ty = rty[r];
              if (ty == kRowBox) { ssum += RowLim[16 * m + M::rowof(g, r)] * fmax(v2_, 0.0); if (v2_ < -vtol) bad = 1; }
              else if (ty == kRowPeak) {
                if (pk_lane < BIGC) ssum += pk_lane * fmax(v2_, 0.0); else if (v2_ > vtol) bad = 1;
                if (v2_ < -vtol) bad = 1;
              } else if (ty == kRowSocRe) {
                const real vi = dv2[4 * m + ((r + 1) & 3)];
                ssum += RowLim[16 * m + M::rowof(g, r)] * sqrt(v2_ * v2_ + vi * vi);
              } else if (ty == kRowSocIm) {
              } else if (fabs(v2_) > vtol) bad = 1;   // free rows admit no ray
            }
          }
          {   // the lane's session: phi(l) = l cap + sum_t [ub (v_t - l)+ + lb (v_t - l)-] at l = min v, max v, 0
            real lmin = BIGC, lmax = -BIGC;
#pragma unroll
            for (int t = 0; t < TS; ++t)
              if ((swm >> t) & 1u) { lmin = fmin(lmin, dv1[t]); lmax = fmax(lmax, dv1[t]); }
            pl_minmax(lmin, lmax);
            real lam3[3] = {lmin, lmax, 0.0};
            real ph3[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              real l_ = lam3[j];
              if (!eq) l_ = fmax(l_, 0.0);
              real ph = half == 0 ? l_ * scap : 0.0;   // (the cap term once per session)
#pragma unroll
              for (int t = 0; t < TS; ++t)
                if ((swm >> t) & 1u) {
                  const real dv = dv1[t] - l_;
                  ph += ubv[t] * fmax(dv, 0.0) + lbv[t] * fmin(dv, 0.0);
                }
              ph3[j] = ph;
            }
            if constexpr (NPW >= 2) { real z_ = 0; pl_sum2(ph3[0], ph3[1]); pl_sum2(ph3[2], z_); }
            const real best = fmin(fmin(ph3[0], ph3[1]), ph3[2]);
            if (smode != 4 && half == 0) ssum += best;
#pragma unroll
            for (int t = 0; t < TS; ++t)
              if (!((swm >> t) & 1u)) ssum += lbv[t] * dv1[t];   // periods outside the window are pinned to lb (= ub)
          }
          real st1[1] = {wave_sum<real>(ssum)}, bm1[1] = {wave_max<real>(bad)};
          pu_sum(st1, 1);
          pu_max(bm1, 1);
          const real stot = uniform_scalar(st1[0]);
          const real bmax = uniform_scalar(bm1[0]);
          if (bmax == 0.0 && stot < -vtol) { status = 3; done = true; }
        }
      }
      if (!done) {   // snapshot for the next certificate test (single precision, as the twin rounds it)
        if constexpr (NPW >= 2) {
#pragma unroll
          for (int t = 0; t < TS; ++t) sn1[t] = (float)y1[t];
#pragma unroll
          for (int r = 0; r < SR; ++r) sn2[r] = (float)y2[r];
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4*>(Snap + c * 256 + lane * 4) = make_float4((float)y1[4 * c], (float)y1[4 * c + 1], (float)y1[4 * c + 2], (float)y1[4 * c + 3]);
          *reinterpret_cast<float4*>(Snap + 3 * 256 + lane * 4) = make_float4((float)y2[0], (float)y2[1], (float)y2[2], (float)y2[3]);
        }
        have_prev = true;
      }
      const real tiny_ = scalar_const(1e-300);
      const real score = fmax(pri / fmax(eps_p, tiny_), dua / fmax(eps_d, tiny_));
      if (score < scalar_const(kStallGain) * best_score) { best_score = uniform_scalar(score); best_it = it; }
      const bool inacc = inaccurate_ok<real>(pri, dua, npri, ndua, A.eps_abs, A.eps_rel, A.inacc_floor);
      const bool stalled = A.stall_iters > 0 && it - best_it >= A.stall_iters && score <= scalar_const(kStallNear) * best_score;
      bool hand_over = false;
      // hand-over to the polish: after polish_iters iterations -- or (polish_stall > 0: ACNQP_EARLY_HANDOVER=1, off by default)
      // from half of them on once the residual score has not improved by 10 % for polish_stall iterations
      const bool pol_due = A.polish_iters > 0 && (it >= A.polish_iters || (A.polish_stall > 0 && 2 * it >= A.polish_iters && it - best_it >= A.polish_stall));
      if (!done && pass == 0 && pol_due) {
        // rows the polish's Schur system would have: one per tight box / peak row, two per tight disc
        real cnt = 0;
        const real ytol = scalar_const(1e-9) * fmax(1.0, qnorm);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          int rty[4];
          row_types(m, rty);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const real yr = y2[4 * m + r], yi = y2[4 * m + ((r + 1) & 3)];
            const bool disc = rty[r] == kRowSocRe;
            const real mag = disc ? sqrt(yr * yr + yi * yi) : yr;
            const bool counts = (disc | (rty[r] == kRowBox) | (rty[r] == kRowPeak)) & (tc < TS) & (tb + tc < Tm) & (mag > ytol);
            cnt += counts ? (disc ? 2.0 : 1.0) : 0.0;
          }
        }
        real c1[1] = {wave_sum<real>(cnt)};
        pu_sum(c1, 1);
        cnt = uniform_scalar(c1[0]);
        hand_over = cnt + 8.0 <= (real)A.pol_rows;
      }
      if (done) {
      } else if (hand_over) {
        status = kStatusPolish;
        done = true;
      } else if (it >= max_iter_p || stalled) {
        done = true;
        if (inacc) status = 5;
      } else if (adapt_p > 0 && it % adapt_p == 0) {   // (inside the residual check only: once per check_every iterations)
        const real e12_ = scalar_const(1e-12);
        const real sp = pri / fmax(npri, e12_);
        const real sd = dua / fmax(ndua, e12_);
        const real ratio = sqrt(sp / fmax(sd, scalar_const(1e-30)));
        const real tol_eff = A.adapt_tol * (1.0 + (real)n_adapt * (1.0 / kAdaptWiden));
        if (ratio > tol_eff || ratio < 1.0 / tol_eff) {
          ++n_adapt;
          rho = uniform_scalar(fmin(fmax(rho * ratio, scalar_const(1e-6)), scalar_const(1e6)));
          a = sigma + pd + rho;
          inv_a = uniform_scalar(1.0 / a);
          inv_rho = uniform_scalar(1.0 / rho);
          __builtin_amdgcn_wave_barrier();
          if (lane < 16 * MT) RowDj[lane] = rho / (a + rho * RowLam[lane]);
          if (aa_m > 0) {   // the fixed-point map changed: restart the ring from the current (z, y)
            aa_cnt = 0; aa_head = 0; aa_valid = 0; aa_have_prev = false; aa_was = false;
            for (int k = lane; k < AM * AM + AM; k += 64) AaH[k] = 0;
#pragma unroll
            for (int t = 0; t < TS; ++t) { up[t] = z1[t] + y1[t] / rho; cp[t] = 0.f; }
#pragma unroll
            for (int r = 0; r < SR; ++r) { up[TS + r] = z2[r] + y2[r] / rho; cp[TS + r] = 0.f; }
          }
          wave_lds_sync();
        }
      }
    }
    STAMP(7);   // residual check (amortised)
  }
  it_total += it;
  // ---- results of this pass: the feasible iterate z1 is the schedule (kept if it beats the earlier passes) ---------------
  if (pass == 0 || status_rank(status) > status_rank(best_status)) {   // wave-uniform
    best_status = status;
    real ol = 0;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int t = 0; t < TS; ++t)
      if (lane_o < N && tb + t < Tm) {
        A.x[((size_t)b * N + lane_o) * Tm + tb + t] = z1[t];
        ol += (0.5 * pd_user * z1[t] + qv[t]) * z1[t];
      }
    if (A.y_out && (!A.y_for_polish_only || status == kStatusPolish)) {   // site-row multipliers, caller's row order and units
      const real* RS = static_cast<const real*>(A.rowscale);
      const int g_o = lane_o >> 4, t_o = lane_o & 15;
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        const int j = 16 * (r >> 2) + M::rowof(g_o, r & 3);
        const int ja = A.rowabi[j];
        if (ja >= 0 && t_o < TS && tb + t_o < Tm) A.y_out[((size_t)b * A.Mg + ja) * Tm + tb + t_o] = y2[r] * RS[j];
      }
    }
    ol = wave_sum<real>(ol);
    { real o1[1] = {ol}; pu_sum(o1, 1); ol = o1[0]; }
    if (lane == 0 && half == 0) {
      // (a mailbox wait that ran into its bound -- never expected -- leaves the problem UNSET: the host entry then fails the
      //  call loudly instead of returning a schedule built on a partner's stale values)
      A.status[b] = xbroken ? 0 : status;
      A.pri[b] = pri;
      A.dua[b] = dua;
      A.obj[b] = ol;
      if (status == kStatusPolish) A.pol_list[atomicAdd(A.pol_count, 1)] = b;
    }
  }
  if (lane == 0 && half == 0) A.iters[b] = it_total;
#ifdef ACNQP_STAMPS
  {
    unsigned long long st_rt1, st_t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t1)::"memory");
    st_acc[9] = st_rt1 - st_rt0;    // 100 MHz ticks of the solver loop
    st_acc[10] = st_t1 - st_t0;     // s_memtime ticks of the same interval
  }
  if (lane == 0 && b < 1024)
    for (int k = 0; k < 24; ++k) g_stamps[(b * 16) * 12 + k] = st_acc[k];
#endif
  if (!retry_wanted(pass, A.retry_passes, status, it, A.stall_iters, A.adapt_every)) break;
  }   // passes
  }   // work queue
#undef BIGC
}

}  // namespace acnqp
