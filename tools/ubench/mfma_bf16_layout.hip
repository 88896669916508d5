#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// hypothesis: A[i][k]: lane l holds i = l%32, k = 8*(l/32)+e (e=0..7); B[k][j]: lane l holds j = l%32, k = 8*(l/32)+e; C[i][j]: lane l holds j = l%32, reg v: i = 8*(v/4) + 4*(l/32) + v%4
__global__ void k(const float* A, const float* B, float* C) {   // A: 32x16 row-major, B: 16x32 row-major, C: 32x32
  const int l = threadIdx.x, h = l / 32, m = l % 32;
  bf8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)A[m * 16 + 8 * h + e]; b[e] = (__bf16)B[(8 * h + e) * 32 + m]; }
  f16v c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int v = 0; v < 16; v++) C[(8 * (v / 4) + 4 * h + v % 4) * 32 + m] = c[v];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hC[32 * 32], *dA, *dB, *dC;
  for (int i = 0; i < 32 * 16; i++) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double s = 0; for (int k = 0; k < 16; k++) s += (double)hA[i * 16 + k] * hB[k * 32 + j]; maxerr = fmax(maxerr, fabs(s - hC[i * 32 + j])); }
  printf("layout hypothesis max err %g (0 = confirmed)\n", maxerr);
  return 0;
}
