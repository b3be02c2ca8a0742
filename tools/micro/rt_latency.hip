// Round-trip cost of the long-horizon kernel's memory pattern on one CU: a workgroup of W waves, each wave repeating
//   load NL tiles (512 B per wave-load, L2-resident, written by ANOTHER wave the step before) -> one dependent FMA per
//   value -> store NS tiles -> workgroup barrier
// Prints shader cycles per step (s_memtime) for several (W, NL, NS).  hipcc --offload-arch=gfx950 -O3 rt_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NL, int NS>
__global__ void chain(double* buf, int steps, int tiles_per_wave, unsigned long long* out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  double* base = buf + (size_t)blockIdx.x * nw * tiles_per_wave * 64;
  unsigned long long t0, t1;
  double acc = 0;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int s = 0; s < steps; ++s) {
    const int src = (wave + s) % nw, dst = (wave + s + 1) % nw;     // read what a neighbour wrote last step
    double v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = base[(size_t)(src * tiles_per_wave + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < NL; ++k) acc = acc * 0.999 + v[k];
#pragma unroll
    for (int k = 0; k < NS; ++k) base[(size_t)(dst * tiles_per_wave + k) * 64 + lane] = acc + k;
    __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) buf[0] = acc;
}
// the same step with 16 bytes per lane (double2): half the instructions for the same bytes
template <int NL, int NS>
__global__ void chain2(double2* buf, int steps, int tiles_per_wave, unsigned long long* out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  double2* base = buf + (size_t)blockIdx.x * nw * tiles_per_wave * 64;
  unsigned long long t0, t1;
  double acc = 0;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int s = 0; s < steps; ++s) {
    const int src = (wave + s) % nw, dst = (wave + s + 1) % nw;
    double2 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = base[(size_t)(src * tiles_per_wave + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < NL; ++k) acc = acc * 0.999 + v[k].x + v[k].y;
#pragma unroll
    for (int k = 0; k < NS; ++k) base[(size_t)(dst * tiles_per_wave + k) * 64 + lane] = make_double2(acc + k, acc - k);
    __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) buf[0].x = acc;
}
template <int NL, int NS>
void run2(int waves, int blocks, double* buf, unsigned long long* dout) {
  const int steps = 2000;
  for (int rep = 0; rep < 2; ++rep) { chain2<NL, NS><<<blocks, waves * 64>>>((double2*)buf, steps, 32, dout); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto x : h) m += (double)x; m /= blocks;
  printf("waves %2d blocks %4d loads %2d stores %2d x 16 B/lane : %8.0f cycles / step\n", waves, blocks, NL, NS, m / steps);
}
template <int NL, int NS>
void run(int waves, int blocks, double* buf, unsigned long long* dout) {
  const int steps = 2000;
  chain<NL, NS><<<blocks, waves * 64>>>(buf, steps, 64, dout);
  hipDeviceSynchronize();
  chain<NL, NS><<<blocks, waves * 64>>>(buf, steps, 64, dout);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto x : h) m += (double)x; m /= blocks;
  printf("waves %2d blocks %4d loads %2d stores %2d : %8.0f cycles / step\n", waves, blocks, NL, NS, m / steps);
}
int main() {
  double* buf; unsigned long long* dout;
  hipMalloc(&buf, (size_t)1024 * 16 * 64 * 64 * 8);
  hipMemset(buf, 0, (size_t)1024 * 16 * 64 * 64 * 8);
  hipMalloc(&dout, 1024 * 8);
  for (int blocks : {1, 16, 256}) {
    for (int w : {4, 8, 16}) {
      run<1, 1>(w, blocks, buf, dout);
      run<8, 8>(w, blocks, buf, dout);
      run<27, 27>(w, blocks, buf, dout);
      run<27, 0>(w, blocks, buf, dout);
      run<1, 27>(w, blocks, buf, dout);
      run2<14, 14>(w, blocks, buf, dout);   // the bytes of <27, 27> (rounded up) in half the instructions
      run2<14, 0>(w, blocks, buf, dout);
    }
  }
  return 0;
}
