#include <hip/hip_runtime.h>
#include <cstdio>
#include "acn_qp_stream.hpp"
int main() {
  for (int v = 0; v < 3; ++v) {
    int nb = -1;
    size_t lds; hipError_t e;
    if (v == 0) { acnqp::StreamLds L(3, 3); lds = L.total * 8; auto k = &acnqp::admm_stream_kernel<3, 3>; if (lds > 65536) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, acnqp::kStreamWaves * 64, lds); hipFuncAttributes a; hipFuncGetAttributes(&a, (const void*)k); printf("<3,3> lds %zu regs %d scratch %zu static lds %zu: ", lds, a.numRegs, a.localSizeBytes, a.sharedSizeBytes); }
    if (v == 1) { acnqp::StreamLds L(2, 1); lds = L.total * 8; auto k = &acnqp::admm_stream_kernel<1, 2>; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, acnqp::kStreamWaves * 64, lds); printf("<1,2> lds %zu: ", lds); }
    if (v == 2) { acnqp::StreamLds L(1, 3); lds = L.total * 8; auto k = &acnqp::admm_stream_kernel<3, 1>; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, acnqp::kStreamWaves * 64, lds); printf("<3,1> lds %zu: ", lds); }
    printf("max active blocks per CU = %d (%s)\n", nb, hipGetErrorString(e));
  }
  return 0;
}
