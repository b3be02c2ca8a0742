// Register-resident kernel, one column tile (horizons <= 16): instantiations and launcher.
#include "acn_qp_tiled_launch.hpp"

namespace acnqp {

hipError_t launch_tiled_ct1(const TiledArgs& a, hipStream_t st) {
  switch (a.MR / 16) {
    case 1: return launch_k<4, 1, 1>(a, st);
    case 2: return launch_k<4, 1, 2>(a, st);
    default: return launch_k<4, 1, 3>(a, st);
  }
}

}  // namespace acnqp

#ifdef ACNQP_STAMPS
/* diagnostic build only: copy the per-phase cycle counters of THIS translation unit's kernels (the headline shapes:
 * one column tile) to the host -- every unit that includes acn_qp_tiled.hpp has its own g_stamps */
extern "C" int acnqp_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(acnqp::g_stamps), sizeof(unsigned long long) * n);
}
#endif
