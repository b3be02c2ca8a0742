// Register-resident kernel, two column tiles (horizons 17 ... 32): instantiations and launcher.
#include "acn_qp_tiled_launch.hpp"

namespace acnqp {

hipError_t launch_tiled_ct2(const TiledArgs& a, hipStream_t st) {
  // two column tiles x one row tile only: with two or three row tiles every wave carries the whole site-row state
  // redundantly and the kernel spills 430-1,100 registers; those shapes run through the long-horizon kernel (LDS-resident
  // where it fits) or the general-shape kernel (acn_qp_api.hip, tiled_shape)
  if (a.MR != 16) return hipErrorInvalidValue;
  return launch_k<4, 2, 1>(a, st);
}

}  // namespace acnqp
