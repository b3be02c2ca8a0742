// Verifies the lane maps of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 assumed by the
// tiled kernel: A[row=l&15][k=l>>4], B[k=l>>4][col=l&15]; D f64: row=(l>>4)+4*reg, f32: row=4*(l>>4)+reg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k64(const double* A, const double* B, double* D) {  // A 16x16 (K=16), B 16x16, row-major
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  d4 acc = {0, 0, 0, 0};
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c * 16 + 4 * s + g], B[(4 * s + g) * 16 + c], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + c] = acc[r];
}
__global__ void k32(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  f4 acc = {0, 0, 0, 0};
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * 16 + 4 * s + g], B[(4 * s + g) * 16 + c], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}
// chaining test: E = A2 * (A*B) using the accumulator registers directly as B operand (reg s <-> k-step s)
__global__ void chain64(const double* A, const double* B, const double* A2, double* E) {
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  d4 acc = {0, 0, 0, 0};
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c * 16 + 4 * s + g], B[(4 * s + g) * 16 + c], acc, 0, 0, 0);
  d4 e = {0, 0, 0, 0};
  // acc[r] is row (g + 4r) of A*B at column c; k-step s needs row 4s+g' ... with slot g: row_of(g, s) = g + 4s
  for (int s = 0; s < 4; ++s) e = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[c * 16 + (g + 4 * s)], acc[s], e, 0, 0, 0);
  for (int r = 0; r < 4; ++r) E[(g + 4 * r) * 16 + c] = e[r];
}
int main() {
  double hA[256], hB[256], hA2[256], hD[256], hE[256], rD[256], rE[256];
  float fA[256], fB[256], fD[256];
  for (int i = 0; i < 256; ++i) { hA[i] = (i * 7 % 13) - 6; hB[i] = (i * 5 % 11) - 5 + (i / 16) * 0.5; hA2[i] = (i * 3 % 7) - 3; fA[i] = hA[i]; fB[i] = hB[i]; }
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double s = 0; for (int k = 0; k < 16; ++k) s += hA[r * 16 + k] * hB[k * 16 + c]; rD[r * 16 + c] = s; }
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double s = 0; for (int k = 0; k < 16; ++k) s += hA2[r * 16 + k] * rD[k * 16 + c]; rE[r * 16 + c] = s; }
  double *dA, *dB, *dA2, *dD, *dE; float *gA, *gB, *gD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dA2, 2048); hipMalloc(&dD, 2048); hipMalloc(&dE, 2048);
  hipMalloc(&gA, 1024); hipMalloc(&gB, 1024); hipMalloc(&gD, 1024);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dA2, hA2, 2048, hipMemcpyHostToDevice);
  hipMemcpy(gA, fA, 1024, hipMemcpyHostToDevice); hipMemcpy(gB, fB, 1024, hipMemcpyHostToDevice);
  k64<<<1, 64>>>(dA, dB, dD); k32<<<1, 64>>>(gA, gB, gD); chain64<<<1, 64>>>(dA, dB, dA2, dE);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(fD, gD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hE, dE, 2048, hipMemcpyDeviceToHost);
  double e64 = 0, e32 = 0, ec = 0;
  for (int i = 0; i < 256; ++i) { e64 = fmax(e64, fabs(hD[i] - rD[i])); e32 = fmax(e32, fabs(fD[i] - rD[i])); ec = fmax(ec, fabs(hE[i] - rE[i])); }
  printf("mfma f64 max err %.3g   f32 max err %.3g   chained f64 max err %.3g\n", e64, e32, ec);
  return (e64 > 1e-9 || e32 > 1e-3 || ec > 1e-9) ? 1 : 0;
}
