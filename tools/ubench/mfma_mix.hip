// Decision microbenchmark for a hybrid kernel: does one fp64 MFMA (16x16x4, producing r^2 for 4 pairs per lane) per
// 4 x {v_rsq_f64 + 7 fp64 VALU ops} hide behind the VALU work of the OTHER waves on the SIMD?
// Compares, per loop iteration of one wave (4 pairs per lane):
//   A: VALU only, exact geometry:      4 x (6 + rsq + 6)   (today's kernel: 12 fp64 ops + rsq per pair)
//   B: VALU only, without geometry:    4 x (rsq + 7)       (lower bound if r^2 were free)
//   C: 1 MFMA + 4 x (rsq + 7)          (hybrid)
// at 4 and 8 waves per SIMD.  Standalone tool (see valu_rates.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

#define FMA(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define RSQ(a) asm volatile("v_rsq_f64 %0, %0" : "+v"(a));
#define PAIR_TAIL(a) RSQ(a) FMA(a) FMA(a) FMA(a) FMA(a) FMA(a) FMA(a) FMA(a)
#define GEOM(a) FMA(a) FMA(a) FMA(a) FMA(a) FMA(a)   /* 6 geometry ops - 1 (the tail has 7 instead of 6) */

template <int MODE> __global__ void __launch_bounds__(512) k(double* out, int iters) {
  double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3, b = 1.0000001, c = 1e-9;
  double4_t acc = {0, 0, 0, 0}, cc = {1, 2, 3, 4};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (MODE == 2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, cc, 0, 0, 0);
      if (MODE == 0) { GEOM(a0) GEOM(a1) GEOM(a2) GEOM(a3) }
      PAIR_TAIL(a0) PAIR_TAIL(a1) PAIR_TAIL(a2) PAIR_TAIL(a3)
      if (MODE == 2) { a0 += acc[0] * 1e-30; a1 += acc[1] * 1e-30; a2 += acc[2] * 1e-30; a3 += acc[3] * 1e-30; }   // consume D (4 extra fma)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + acc[0];
}

int main() {
  CHECK(hipSetDevice(0));
  const char* names[3] = {"A valu exact (12+rsq per pair)", "B valu w/o geometry (7+rsq)", "C mfma + (8+rsq)"};
  void (*fns[3])(double*, int) = {k<0>, k<1>, k<2>};
  for (int wps : {2, 4, 8}) {
    for (int m = 0; m < 3; m++) {
      const int threads = 512, nblk = 256 * wps / 2;   // 8 waves per block = 2 per SIMD
      double* out; CHECK(hipMalloc(&out, sizeof(double) * nblk * threads));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      const int iters = 20000;
      hipLaunchKernelGGL(fns[m], dim3(nblk), dim3(threads), 0, 0, out, 200); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(fns[m], dim3(nblk), dim3(threads), 0, 0, out, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      // pairs per SIMD: wps waves x iters x 4 unroll x 4 pairs per lane-step
      const double pair_steps = (double)wps * iters * 16;
      printf("waves/SIMD=%d  %-34s %8.2f ms  %.2f ns per pair-step per SIMD  (x2.15 GHz = %.1f cycles)  -> %.2e pairs/s chip\n", wps, names[m], ms,
             ms * 1e6 / pair_steps, ms * 1e6 / pair_steps * 2.15, 1024.0 * 64 * pair_steps / (ms * 1e-3));
      CHECK(hipFree(out));
    }
  }
  return 0;
}
