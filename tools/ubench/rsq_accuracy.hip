// Accuracy of the hardware reciprocal-square-root seeds on gfx950 and the steady-state
// shader clock under a dense fp64 VALU load.  Standalone tool (see valu_rates.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_rsq(const double* x, double* y64, float* y32, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i], r; float af = (float)a, rf;
  asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "v"(a));
  asm volatile("v_rsq_f32 %0, %1" : "=v"(rf) : "v"(af));
  y64[i] = r; y32[i] = rf;
}
// long dense fp64 FMA loop, stamps shader clock (s_memtime) and the 100 MHz constant clock (s_memrealtime)
__global__ void __launch_bounds__(256) k_clock(double* out, int iters, long long* stamps, int nfma_kind) {
  double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3, a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;
  double b = 1.0000001, c = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#define F(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
      F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
  CHECK(hipSetDevice(0));
  const int n = 1 << 22;
  std::vector<double> x(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(1.0, 4.0), E(-300, 300);
  for (int i = 0; i < n; i++) x[i] = (i < n / 2) ? U(g) : U(g) * std::pow(2.0, std::floor(E(g)));
  double *dx, *dy; float* dyf;
  CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dy, n * 8)); CHECK(hipMalloc(&dyf, n * 4));
  CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, dx, dy, dyf, n);
  CHECK(hipDeviceSynchronize());
  std::vector<double> y(n); std::vector<float> yf(n);
  CHECK(hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(yf.data(), dyf, n * 4, hipMemcpyDeviceToHost));
  double e64 = 0, e32 = 0, e64w = 0;
  for (int i = 0; i < n; i++) {
    long double ex = 1.0L / sqrtl((long double)x[i]);
    double r = fabs((double)((y[i] - ex) / ex));
    if (i < n / 2) { e64 = fmax(e64, r); long double exf = 1.0L / sqrtl((long double)(float)x[i]); e32 = fmax(e32, fabs((double)((yf[i] - exf) / exf))); }
    else e64w = fmax(e64w, r);
  }
  printf("v_rsq_f64 max rel err on [1,4): %.3e (2^%.2f); wide exponent range: %.3e (2^%.2f)\n", e64, log2(e64), e64w, log2(e64w));
  printf("v_rsq_f32 max rel err on [1,4): %.3e (2^%.2f)\n", e32, log2(e32));
  // special values
  double sp[4] = {0.0, 1e-320, INFINITY, -1.0};
  CHECK(hipMemcpy(dx, sp, 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rsq, dim3(1), dim3(256), 0, 0, dx, dy, dyf, 4);
  CHECK(hipMemcpy(y.data(), dy, 32, hipMemcpyDeviceToHost));
  printf("v_rsq_f64(0)=%g (1e-320)=%g (inf)=%g (-1)=%g\n", y[0], y[1], y[2], y[3]);
  // steady-state clock
  for (int wps : {1, 2, 4, 8}) {
    const int nblk = 256 * wps; double* out; long long* st;
    CHECK(hipMalloc(&out, nblk * 256 * 8)); CHECK(hipMalloc(&st, nblk * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      const int iters = rep == 0 ? 20000 : 400000 / wps;     // second launch: ~0.1-0.5 s
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_clock, dim3(nblk), dim3(256), 0, 0, out, iters, st, 0);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(nblk * 2); CHECK(hipMemcpy(h.data(), st, nblk * 16, hipMemcpyDeviceToHost));
      double clk = 0; for (int b = 0; b < nblk; b++) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; clk /= nblk;
      const double ninstr = (double)iters * 128 * wps;   // per SIMD
      printf("fp64 FMA loop, %d blocks of 256 (%d waves/SIMD), %8.2f ms: in-kernel clock %.3f GHz, %.3f ns per wave-instr per SIMD = %.2f cycles, %.1f TFLOP/s\n",
             nblk, wps, ms, clk, ms * 1e6 / ninstr, ms * 1e6 / ninstr * clk, 128.0 * ninstr * 1024 / (ms * 1e-3) * 1e-12);
    }
    CHECK(hipFree(out)); CHECK(hipFree(st));
  }
  return 0;
}
