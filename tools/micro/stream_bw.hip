// Achievable HBM bandwidth of the large-site kernel's access pattern on this GPU: read-modify-write of 512-byte
// (8 B per lane) or 1-KB (16 B per lane) wave rows, temporal or non-temporal, 2 reads : 1 write like the solver
// (reads a, b; writes a).  Prints GB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W, bool NT> __global__ void rmw(double* __restrict__ a, const double* __restrict__ b, size_t n_rows) {
  // each wave walks rows; a row = 64 lanes x W doubles
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nw = (size_t)gridDim.x * (blockDim.x >> 6);
  for (size_t r = wave; r < n_rows; r += nw) {
    double* pa = a + (r * 64 + lane) * W;
    const double* pb = b + (r * 64 + lane) * W;
    double va[W], vb[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { va[k] = NT ? __builtin_nontemporal_load(pa + k) : pa[k]; vb[k] = NT ? __builtin_nontemporal_load(pb + k) : pb[k]; }
#pragma unroll
    for (int k = 0; k < W; ++k) { const double v = va[k] * 1.0000001 + vb[k]; if (NT) __builtin_nontemporal_store(v, pa + k); else pa[k] = v; }
  }
}
template <int W, bool NT> void run(const char* name, double* a, double* b, size_t n, int blocks, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t rows = n / (64 * W);
  rmw<W, NT><<<blocks, threads>>>(a, b, rows); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) rmw<W, NT><<<blocks, threads>>>(a, b, rows); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-34s blocks %5d x %4d: %.3f ms -> %.0f GB/s (2 reads + 1 write of %.2f GiB arrays)\n", name, blocks, threads, ms, 3.0 * n * 8 / ms / 1e6, n * 8.0 / (1 << 30));
}
int main() {
  const size_t n = (size_t)3 << 27;   // 3 GiB per array
  double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  for (int threads : {256, 512}) for (int blocks : {256, 512, 2048}) {
    run<1, false>("8 B/lane", a, b, n, blocks, threads);
    run<1, true>("8 B/lane non-temporal", a, b, n, blocks, threads);
    run<2, false>("16 B/lane", a, b, n, blocks, threads);
    run<2, true>("16 B/lane non-temporal", a, b, n, blocks, threads);
  }
  return 0;
}
