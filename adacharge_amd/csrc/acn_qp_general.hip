// General-shape kernel (acn_qp_general.hpp): instantiations and launcher.
#include "acn_qp_launch.hpp"

namespace acnqp {

// workgroup size by problem size: the plain loops are latency-bound, more threads per problem hide more of it
hipError_t launch_general(const GeneralArgs& ga, int threads, hipStream_t st) {
  if (threads == 256) hipLaunchKernelGGL((admm_general_kernel<double, 256>), dim3(launch_grid(&admm_general_kernel<double, 256>, 256, 0, ga.t)), dim3(256), 0, st, ga);
  else if (threads == 512) hipLaunchKernelGGL((admm_general_kernel<double, 512>), dim3(launch_grid(&admm_general_kernel<double, 512>, 512, 0, ga.t)), dim3(512), 0, st, ga);
  else hipLaunchKernelGGL((admm_general_kernel<double, 1024>), dim3(launch_grid(&admm_general_kernel<double, 1024>, 1024, 0, ga.t)), dim3(1024), 0, st, ga);
  return hipGetLastError();
}

}  // namespace acnqp
