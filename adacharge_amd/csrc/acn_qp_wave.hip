// Wave-per-problem kernel (acn_qp_wave.hpp): instantiation and launcher.
#include "acn_qp_launch.hpp"
#include "acn_qp_wave.hpp"

namespace acnqp {

// shapes the wave-per-problem kernel takes: a lane per EVSE, twelve period registers, one session slot, one row tile,
// box / disc / peak rows and the two prox rows.  A function of the SHAPE only, never of the batch size: a problem's result does not depend on
// what it is batched with (tests/test_gpu_parity.py asserts the bits).  The price: a problem is ONE wave's dependent chain
// here (4.3 us per iteration) and four waves' in the tiled kernel (3.1 us alone on a CU), so a launch of at most one
// problem per CU ends later than it did (256 problems: 2.9 against 2.3 ms; one problem: 0.85 against 0.6 ms) -- from two
// problems per CU on, four problems in flight per CU win (16,384: 26.5 -> 15.5 ms).  ACNQP_WAVE_MIN_BATCH=n (diagnostic)
// sends launches of fewer than n problems to the tiled kernel.  Returns the variant (0: not this kernel; 1: horizon
// <= 12, one wave per problem; 2: horizon 13 ... 24, two waves; 3: two row tiles at horizon <= 12, two waves of six periods;
// 4: two row tiles at horizon 13 ... 24, four waves of six periods -- one problem per workgroup; 5: one row tile at
// horizon 33 ... 48, four waves of twelve periods).
int wave_shape(int N, int t_max, int k_sessions, int MR, bool has_max, int batch) {
  static const bool off = std::getenv("ACNQP_NO_WAVE") != nullptr;   // diagnostic / A-B: the register-resident tiled kernel instead
  static const bool off2 = std::getenv("ACNQP_NO_WAVE2") != nullptr; // ... for the two-waves-per-problem variants only
  static const int min_batch = std::getenv("ACNQP_WAVE_MIN_BATCH") ? std::atoi(std::getenv("ACNQP_WAVE_MIN_BATCH")) : 1;
  (void)has_max;   // (the demand-charge row's prox couples all periods: its sums cross the group's mailbox)
  if (off || N > 64 || k_sessions != 1 || batch < min_batch) return 0;
  if (MR == 16 && t_max <= kWaveTS) return 1;                   // one wave per problem
  if (MR == 16 && t_max <= 2 * kWaveTS) return off2 ? 0 : 2;    // two waves, twelve periods each
  // (horizons 25 ... 32 stay with the tiled kernel's two column tiles: the same four waves per problem there, 19.8 against
  //  21.3 ms at 2,048 problems; from 33 on the alternative streams its state: 57.0 against 22.3 ms at horizon 48)
  if (MR == 16 && t_max > 32 && t_max <= 4 * kWaveTS) return off2 ? 0 : 5;   // horizon 33 ... 48: four waves, twelve periods each
  if (MR == 32 && t_max <= kWaveTS) return off2 ? 0 : 3;        // two row tiles: two waves, six periods each
  if (MR == 32 && t_max <= 2 * kWaveTS) return off2 ? 0 : 4;    // two row tiles, horizon 13 ... 24: four waves, six periods each
  return 0;
}

template <int NPW, int TSV, int MT, bool PROX>
static hipError_t launch_wave_prox(const TiledArgs& a_in, hipStream_t st) {
  TiledArgs a = a_in;
  a.accel_mem = std::min(a.accel_mem, kWaveAM);
  const WaveLds L(a.accel_mem, NPW, MT, TSV);
  const size_t lds = (size_t)L.total * 8;
  auto kern = &admm_wave_kernel<kWaveAM, NPW, TSV, MT, PROX>;
  if (lds > 64 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  // one queue position per WAVE (pair of waves): a workgroup serves kWaveNW / NPW positions at a time
  constexpr int per_wg = kWaveNW / NPW;
  const int groups = (a.B + per_wg - 1) / per_wg;
  int grid = groups;
  if (a.queue) {
    const int per_cu = resident_per_cu(reinterpret_cast<const void*>(kern), kWaveNW * 64, lds), cus = device_cus();
    const int cap = a.grid_cap > 0 ? a.grid_cap : groups;   // (a cap on WORKGROUPS: the resume launches of pipelined chunks keep theirs small)
    // A wave takes its problems one after the other, so a workgroup is as slow as the slowest of its four waves' shares:
    // only PERSISTENT workgroups level that (a workgroup per four problems wastes max-of-four against mean-of-four, 40 %).
    // The pipelined host entries (grid_oversub > 1 for the other kernels) therefore keep this kernel's grid at the resident
    // slots too; the next stream's launch moves in as this one's workgroups retire (measured, 16,384 problems per step:
    // half the chip per launch 21.4-21.8 ms, the whole chip 20.0-20.5 ms; tools/sweep_wave_pipeline.sh).
    int resident = per_cu * cus;
    static const int grid_env = std::getenv("ACNQP_WAVE_GRID") ? std::atoi(std::getenv("ACNQP_WAVE_GRID")) : 0;   // diagnostic: workgroups per pipelined launch
    if (a.grid_oversub > 1 && grid_env > 0) resident = grid_env;
    grid = std::max(1, std::min(std::min(groups, cap), resident));
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaveNW * 64), lds, st, a);
  return hipGetLastError();
}

// (sites without a prox row run the instantiation that carries no prox code)
template <int NPW, int TSV, int MT>
static hipError_t launch_wave_npw(const TiledArgs& a, hipStream_t st) {
  return a.lf != nullptr || a.dc != nullptr ? launch_wave_prox<NPW, TSV, MT, true>(a, st) : launch_wave_prox<NPW, TSV, MT, false>(a, st);
}

int wave_accel_columns() { return kWaveAM; }

hipError_t launch_wave(const TiledArgs& a, hipStream_t st) {
  if (a.MR == 32) return a.Tm <= kWaveTS ? launch_wave_npw<2, 6, 2>(a, st) : launch_wave_npw<4, 6, 2>(a, st);
  if (a.Tm > 2 * kWaveTS) return launch_wave_npw<4, 12, 1>(a, st);
  return a.Tm <= kWaveTS ? launch_wave_npw<1, 12, 1>(a, st) : launch_wave_npw<2, 12, 1>(a, st);
}

}  // namespace acnqp

#ifdef ACNQP_STAMPS
/* diagnostic build only: the per-phase cycle counters of the wave-per-problem kernel (this unit's own g_stamps) */
extern "C" int acnqp_debug_read_wave_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(acnqp::g_stamps), sizeof(unsigned long long) * n);
}
#endif
