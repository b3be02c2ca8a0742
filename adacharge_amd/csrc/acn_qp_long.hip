// Long-horizon kernel (acn_qp_long.hpp): instantiations and launcher.
#include <cstdlib>

#include "acn_qp_launch.hpp"

namespace acnqp {

int long_tiles(int t_max) { return t_max <= 32 ? 2 : (t_max <= 96 ? 6 : (t_max <= 144 ? 9 : 18)); }

template <int CTL, int MT>
static hipError_t launch_long_one(const StreamArgs& sa, hipStream_t st) {
  // 8 waves per problem: 256 registers per lane hold a row item of any supported horizon without scratch (16 waves
  // at 128 registers spilled ~230 of them and ran slower)
  constexpr int NWV = 8;
  const size_t lds = (size_t)2 * MT * CTL * 256 * sizeof(double);   // e^, h^
  auto kern = &admm_long_kernel<CTL, MT, NWV>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NWV * 64, lds, sa.t)), dim3(NWV * 64), lds, st, sa);
  return hipGetLastError();
}

// r0 / zh in LDS when the array fits next to e^, h^ (acn_qp_long.hpp, RZL): <= 156 KB of dynamic LDS
template <int CTL, int MT>
static hipError_t launch_long_rzl(const StreamArgs& sa, hipStream_t st) {
  constexpr int NWV = 8;
  const int NE = sa.t.NP / 16;
  const size_t lds = (size_t)256 * CTL * (2 * MT + NE) * sizeof(double);
  auto kern = &admm_long_kernel<CTL, MT, NWV, false, true>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NWV * 64, lds, sa.t)), dim3(NWV * 64), lds, st, sa);
  return hipGetLastError();
}
// ... and x as well (horizons up to 96: 2 x 48 KB next to e^, h^)
template <int MT>
static hipError_t launch_long_xsl(const StreamArgs& sa, hipStream_t st) {
  constexpr int NWV = 8, CTL = 6;
  const int NE = sa.t.NP / 16;
  const size_t lds = (size_t)256 * CTL * (2 * MT + 2 * NE) * sizeof(double);
  auto kern = &admm_long_kernel<CTL, MT, NWV, false, true, true>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NWV * 64, lds, sa.t)), dim3(NWV * 64), lds, st, sa);
  return hipGetLastError();
}
static bool xsl_fits(int CTL, int MT, int NP) {
  static const bool off = std::getenv("ACNQP_NO_XSL") != nullptr || std::getenv("ACNQP_NO_RZL") != nullptr;   // diagnostics
  return !off && CTL == 6 && (size_t)256 * CTL * (2 * MT + 2 * (NP / 16)) * sizeof(double) <= (size_t)156 * 1024;
}
static bool rzl_fits(int CTL, int MT, int NP) {
  static const bool off = std::getenv("ACNQP_NO_RZL") != nullptr;   // diagnostic: r0 / zh in the workspace
  return !off && (CTL == 6 || CTL == 9) && (size_t)256 * CTL * (2 * MT + NP / 16) * sizeof(double) <= (size_t)156 * 1024;
}

template <int CTL, int MT>
static hipError_t launch_long_lds(const StreamArgs& sa, hipStream_t st) {
  constexpr int NWV = 8;
  const int NE = sa.t.NP / 16;
  const size_t lds = (size_t)256 * CTL * (5 * MT + 7 * NE) * sizeof(double);   // e^, h^, site rows; 7 iterate arrays
  auto kern = &admm_long_kernel<CTL, MT, NWV, true>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NWV * 64, lds, sa.t)), dim3(NWV * 64), lds, st, sa);
  return hipGetLastError();
}

hipError_t launch_long(const StreamArgs& sa, hipStream_t st, bool lds_resident) {
  const int CTL = long_tiles(sa.t.Tm), MT = sa.t.MR / 16;
  if (lds_resident) return launch_long_lds<2, 2>(sa, st);   // lds_long_shape: two column tiles, two row tiles
  if (xsl_fits(CTL, MT, sa.t.NP)) return MT == 1 ? launch_long_xsl<1>(sa, st) : launch_long_xsl<2>(sa, st);
  if (rzl_fits(CTL, MT, sa.t.NP)) {
    switch (CTL * 10 + MT) {
      case 61: return launch_long_rzl<6, 1>(sa, st);
      case 62: return launch_long_rzl<6, 2>(sa, st);
      case 91: return launch_long_rzl<9, 1>(sa, st);
      default: return launch_long_rzl<9, 2>(sa, st);
    }
  }
  switch (CTL * 10 + MT) {
    case 21: return launch_long_one<2, 1>(sa, st);
    case 22: return launch_long_one<2, 2>(sa, st);
    case 61: return launch_long_one<6, 1>(sa, st);
    case 62: return launch_long_one<6, 2>(sa, st);
    case 91: return launch_long_one<9, 1>(sa, st);
    case 92: return launch_long_one<9, 2>(sa, st);
    case 181: return launch_long_one<18, 1>(sa, st);
    default: return launch_long_one<18, 2>(sa, st);
  }
}

}  // namespace acnqp

#ifdef ACNQP_STAMPS
/* diagnostic build only: the long-horizon kernel's per-phase cycle counters (this unit's copy of g_stamps) */
extern "C" int acnqp_debug_read_stamps_long(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(acnqp::g_stamps), sizeof(unsigned long long) * n);
}
#endif
