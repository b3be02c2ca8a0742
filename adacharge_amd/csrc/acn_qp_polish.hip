// Polish kernel (acn_qp_polish.hpp): instantiation and launcher.
#include "acn_qp_launch.hpp"

namespace acnqp {

int polish_rows_that_fit(int N, int Tm, int Mg, int nrow) {
  return PolishLds::rows_that_fit(N, Tm, Mg, nrow, kLdsPerCu - 2048);
}

hipError_t launch_polish(const PolishArgs& pa, int cus, hipStream_t st) {
  const int nrow = pa.M + (pa.has_peak ? 1 : 0);
  const PolishLds L(pa.N, pa.Tm, pa.Mg, nrow, pa.max_rows);
  const size_t lds = (size_t)L.total * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polish_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // one workgroup per CU at most (the Schur system fills the LDS); the workgroups share the list through the queue
  hipLaunchKernelGGL(polish_kernel<0>, dim3(std::max(1, std::min(pa.B, cus))), dim3(kPolThreads), lds, st, pa);
  return hipGetLastError();
}

}  // namespace acnqp
