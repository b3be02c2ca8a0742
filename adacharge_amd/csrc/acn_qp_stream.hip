// Large-site kernel (acn_qp_stream.hpp): instantiations and launcher.
#include "acn_qp_launch.hpp"

namespace acnqp {

template <int CT, int MT, int NWV>
static hipError_t launch_stream_nwv(const StreamArgs& sa, hipStream_t st) {
  const StreamLds L(MT, CT, NWV);
  const size_t lds = (size_t)L.total * sizeof(double);
  auto kern = &admm_stream_kernel<CT, MT, NWV>;
  if (lds > 64 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NWV * 64, lds, sa.t)), dim3(NWV * 64), lds, st, sa);
  return hipGetLastError();
}

// 4 waves per problem and two problems per CU when the batch can fill the chip twice over (throughput), 8 waves per
// problem otherwise (latency); same bits either way (acn_qp_stream.hpp)
template <int CT, int MT>
static hipError_t launch_stream_one(const StreamArgs& sa, hipStream_t st) {
  return sa.t.B >= 384 ? launch_stream_nwv<CT, MT, 4>(sa, st) : launch_stream_nwv<CT, MT, 8>(sa, st);
}

hipError_t launch_stream(const StreamArgs& sa, hipStream_t st) {
  const int CT = (sa.t.Tm + 15) / 16, MT = sa.t.MR / 16;
  switch (CT * 10 + MT) {
    case 11: return launch_stream_one<1, 1>(sa, st);
    case 12: return launch_stream_one<1, 2>(sa, st);
    case 13: return launch_stream_one<1, 3>(sa, st);
    case 21: return launch_stream_one<2, 1>(sa, st);
    case 22: return launch_stream_one<2, 2>(sa, st);
    case 23: return launch_stream_one<2, 3>(sa, st);
    case 31: return launch_stream_one<3, 1>(sa, st);
    case 32: return launch_stream_one<3, 2>(sa, st);
    default: return launch_stream_one<3, 3>(sa, st);
  }
}

}  // namespace acnqp

#ifdef ACNQP_STAMPS
/* diagnostic build only: the large-site kernel's per-phase cycle counters (this unit's copy of g_stamps) */
extern "C" int acnqp_debug_read_stamps_stream(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(acnqp::g_stamps), sizeof(unsigned long long) * n);
}
#endif
