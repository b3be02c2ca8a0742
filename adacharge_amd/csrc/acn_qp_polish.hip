// Polish kernel (acn_qp_polish.hpp): instantiation and launcher.
#include "acn_qp_launch.hpp"

namespace acnqp {

int polish_max_rows(int nrow, int Tm) { return std::min(kPolMaxRows, ((2 * nrow * Tm + 7) / 8) * 8); }

int polish_max_sess(int N, int K) { return std::min(kPolMaxSess, ((N * K + 7) / 8) * 8); }

int polish_blocks_that_fit(int N, int Tm, int Mg, int nrow, int max_sess, int lds_bytes) {
  return PolishLds::blocks_that_fit(N, Tm, Mg, nrow, polish_max_rows(nrow, Tm), max_sess, std::min(lds_bytes, kLdsPerCu - 2048));
}

hipError_t launch_polish(const PolishArgs& pa, int max_grid, hipStream_t st) {
  const int nrow = pa.M + (pa.has_peak ? 1 : 0);
  const PolishLds L(pa.N, pa.Tm, pa.Mg, nrow, pa.max_rows, pa.blk_doubles, pa.max_sess);
  const size_t lds = (size_t)L.total * sizeof(double);
  const bool small = pa.Tm <= 16;
  hipError_t e = small ? ensure_dynamic_lds(reinterpret_cast<const void*>(&polish_kernel<4>), lds)
                       : ensure_dynamic_lds(reinterpret_cast<const void*>(&polish_kernel<8>), lds);
  if (e != hipSuccess) return e;
  // `max_grid` workgroups share the list through the queue (a lone launch: one per CU; a pipelined one: a few -- its list is short)
  const dim3 grid(std::max(1, std::min(pa.B, max_grid)));
  if (small) hipLaunchKernelGGL(polish_kernel<4>, grid, dim3(kPolThreads), lds, st, pa);
  else hipLaunchKernelGGL(polish_kernel<8>, grid, dim3(kPolThreads), lds, st, pa);
  return hipGetLastError();
}

}  // namespace acnqp
