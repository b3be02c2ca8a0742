// Register-resident kernel, two column tiles (horizons 17 ... 32): instantiations and launcher.
#include "acn_qp_tiled_launch.hpp"

namespace acnqp {

hipError_t launch_tiled_ct2(const TiledArgs& a, hipStream_t st) {
  switch (a.MR / 16) {
    case 1: return launch_k<4, 2, 1>(a, st);
    case 2: return launch_k<4, 2, 2>(a, st);
    default: return launch_k<4, 2, 3>(a, st);
  }
}

}  // namespace acnqp
