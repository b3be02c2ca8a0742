// Launch logic of the register-resident kernel, shared by the two translation units that instantiate it
// (one per number of column tiles).
#pragma once
#include "acn_qp_launch.hpp"

namespace acnqp {

template <int NW, int CT, int MT, int KS, int OCC>
hipError_t launch_tiled_occ(const TiledArgs& a, hipStream_t st) {
  constexpr int AM = OCC == 1 ? kAccelMax1 : kAccelMax2;
  const TiledLds L(NW, MT, CT, a.NP, a.K, AM, std::min(a.accel_mem, AM), 8, a.pbuf_single);
  const size_t lds = (size_t)L.total * 8;
  auto kern = &admm_tiled_kernel<double, NW, CT, MT, KS, OCC, AM>;
  if (lds > 64 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(launch_grid(kern, NW * 64, lds, a)), dim3(NW * 64), lds, st, a);
  return hipGetLastError();
}

// Two register budgets of the same kernel.  The 256-register build lets two workgroups share a CU
// (the second hides the first one's dependent-chain latency, and problems of a second batch on another
// stream can move in while stragglers finish); it is taken whenever it exists for the shape and the
// requested Anderson ring fits half the LDS -- a function of the shape only, never of the batch size,
// so that a problem's result does not depend on what it is batched with.
template <int NW, int CT, int MT, int KS>
hipError_t launch_tiled(TiledArgs a, int requested_accel, hipStream_t st) {
  int single = 0;
  const int cap1 = accel_capacity_best(NW, MT, CT, a.NP, a.K, &single);
  a.accel_mem = std::min(requested_accel, cap1);
  a.pbuf_single = 0;
  // one column tile x one row tile x one session slot (the headline shape) is the only one whose five-column ring fits
  // half the LDS; with two row tiles the ring needs the whole CU anyway, so that two-workgroup build (380 spilled
  // registers) was never launched with the default options and is gone
  if constexpr (CT == 1 && MT == 1 && KS == 1) {
    static const bool occ1 = std::getenv("ACNQP_OCC1") != nullptr;   // diagnostic: the one-workgroup-per-CU build
    if (!occ1 && a.accel_mem <= accel_capacity(NW, MT, CT, a.NP, a.K, 2))
      return launch_tiled_occ<NW, CT, MT, KS, 2>(a, st);
  }
  a.pbuf_single = (a.accel_mem > accel_capacity(NW, MT, CT, a.NP, a.K, 1, 0)) ? single : 0;
  return launch_tiled_occ<NW, CT, MT, KS, 1>(a, st);
}

template <int NW, int CT, int MT>
hipError_t launch_k(const TiledArgs& a, hipStream_t st) {
  if (a.K == 1) return launch_tiled<NW, CT, MT, 1>(a, a.accel_mem, st);
  return launch_tiled<NW, CT, MT, kMaxK>(a, a.accel_mem, st);
}

}  // namespace acnqp
