// Host-side launchers of the four kernel families.  Each family is instantiated in its own translation unit
// (acn_qp_tiled_ct1.hip, acn_qp_tiled_ct2.hip, acn_qp_stream.hip, acn_qp_long.hip, acn_qp_general.hip) so that
// adacharge_amd/build.py compiles them in parallel; acn_qp_api.hip (the C ABI) only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "acn_qp_tiled.hpp"
#include "acn_qp_general.hpp"
#include "acn_qp_stream.hpp"
#include "acn_qp_long.hpp"
#include "acn_qp_polish.hpp"

namespace acnqp {

// LDS one workgroup may use: the whole CU when alone, half of it when two share the CU
constexpr int kLdsPerCu = 160 * 1024;
constexpr int kAccelMax1 = 5, kAccelMax2 = 5;   // Anderson columns compiled into the OCC = 1 / OCC = 2 variants

// Anderson columns that fit next to the solver's own LDS for this kernel shape with `occ` workgroups per CU
inline int accel_capacity(int NW, int MT, int CT, int NP, int K, int occ, int pbuf_single = 0) {
  const TiledLds base(NW, MT, CT, NP, K, occ == 1 ? kAccelMax1 : kAccelMax2, 1, 8, pbuf_single);
  const int col = TiledLds::column_bytes(NW, MT, CT);
  const int fixed = base.total * 8 - col;
  const int cap = (kLdsPerCu / occ - fixed - 64) / col;
  return std::max(0, std::min(cap, occ == 1 ? kAccelMax1 : kAccelMax2));
}

// One workgroup per CU: if the double-buffered partial-tile slab leaves fewer than the compiled-in number of ring
// columns, give one slab up (one more barrier per iteration buys a column: on the congested horizon-24 problems a
// fourth / fifth column is worth 5x fewer iterations on the slowest instances)
inline int accel_capacity_best(int NW, int MT, int CT, int NP, int K, int* pbuf_single) {
  const int two = accel_capacity(NW, MT, CT, NP, K, 1, 0);
  const int one = accel_capacity(NW, MT, CT, NP, K, 1, 1);
  *pbuf_single = one > two ? 1 : 0;
  return std::max(one, two);
}

// Grid of a launch.  With the work queue (a.queue set) it is the number of workgroups the chip keeps RESIDENT for this
// kernel -- occupancy x compute units, never more than the problems or than `a.grid_cap` -- and the workgroups fetch
// problems until the queue is empty (queue_next, acn_qp_tiled.hpp); without it one workgroup per problem.
// (occupancy and device queries cost tens of microseconds each: with three launches per pipelined chunk they made the
//  host thread the bottleneck -- 18 chunks per step -- so both are cached per (kernel, block size, LDS) and per device)
inline int resident_per_cu(const void* kern, int threads, size_t lds) {
  struct Key { const void* k; int t; size_t l; int dev; int v; };
  static thread_local Key cache[16];
  static thread_local int used = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  for (int i = 0; i < used; ++i)
    if (cache[i].k == kern && cache[i].t == threads && cache[i].l == lds && cache[i].dev == dev) return cache[i].v;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
  cache[used < 16 ? used++ : 15] = Key{kern, threads, lds, dev, per_cu};
  return per_cu;
}
inline int device_cus() {
  static thread_local int cache[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && cache[dev] > 0) return cache[dev];
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  if (dev >= 0 && dev < 16) cache[dev] = cus;
  return cus;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and growing size, not once per launch
inline hipError_t ensure_dynamic_lds(const void* kern, size_t lds) {
  struct Key { const void* k; int dev; size_t l; };
  static thread_local Key cache[32];
  static thread_local int used = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  for (int i = 0; i < used; ++i)
    if (cache[i].k == kern && cache[i].dev == dev) {
      if (cache[i].l >= lds) return hipSuccess;
      const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) cache[i].l = lds;
      return e;
    }
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess && used < 32) cache[used++] = Key{kern, dev, lds};
  return e;
}
template <typename Kern>
inline int launch_grid(Kern kern, int threads, size_t lds, const TiledArgs& a) {
  if (!a.queue) return a.B;
  const int per_cu = resident_per_cu(reinterpret_cast<const void*>(kern), threads, lds), cus = device_cus();
  return std::max(1, std::min(std::min(a.B, a.grid_cap > 0 ? a.grid_cap : a.B), per_cu * cus * std::max(1, a.grid_oversub)));
}

// register-resident kernel (acn_qp_tiled.hpp): N <= 64, one / two column tiles; a.accel_mem = columns requested
hipError_t launch_tiled_ct1(const TiledArgs& a, hipStream_t st);
hipError_t launch_tiled_ct2(const TiledArgs& a, hipStream_t st);
// wave-per-problem kernel (acn_qp_wave.hpp): N <= 64, one session slot, horizon <= 24 with one or two row tiles or 33 ... 48
// with one; wave_shape says which variant a launch is routed to (0: none; by shape -- `batch` only for the diagnostic
// ACNQP_WAVE_MIN_BATCH)
int wave_shape(int N, int t_max, int k_sessions, int MR, bool has_max, int batch);
hipError_t launch_wave(const TiledArgs& a, hipStream_t st);
int wave_accel_columns();   // Anderson columns compiled into it
// large-site kernel (acn_qp_stream.hpp)
hipError_t launch_stream(const StreamArgs& sa, hipStream_t st);
// long-horizon kernel (acn_qp_long.hpp); lds_resident: its LDS-resident variant for two column tiles x two row tiles
int long_tiles(int t_max);
hipError_t launch_long(const StreamArgs& sa, hipStream_t st, bool lds_resident);
// polish kernel (acn_qp_polish.hpp): rows of the Schur system its LDS holds for a shape (0: does not fit); launch over
// the list the solver kernel left
int polish_blocks_that_fit(int N, int Tm, int Mg, int nrow, int max_sess, int lds_bytes);   // LDS doubles for the per-period blocks (<= the worst case)
int polish_max_sess(int N, int K);                                            // session columns the capacitance matrix holds
int polish_max_rows(int nrow, int Tm);                                        // rows the row tables hold
hipError_t launch_polish(const PolishArgs& pa, int max_grid, hipStream_t st);
// general-shape kernel (acn_qp_general.hpp), `threads` in {256, 512, 1024}
hipError_t launch_general(const GeneralArgs& ga, int threads, hipStream_t st);

}  // namespace acnqp
