// Measured accuracy of the library's three reciprocal-square-root forms (include/sctl_amd/device/ukernels.hpp) against long double:
//   MODE 0: the v_rsq_f64 seed;  MODE 1: the unnormalised Newton step 2/r = y0 (3 - x y0^2), halved here;  MODE 2: the Halley step
//   (kernels whose terms carry several powers of 1/r) and the four-instruction cubic step (8/3)/r = y0 ((x y0^2 - 5/3)^2 + 20/9), divided
//   by its factor here (kernels with one power of 1/r: the Laplace kernels, stresslet, traction).
// With d = the seed's relative error, Newton leaves -3/2 d^2 (always too small) and the cubic steps 5/2 d^3 plus rounding.
// 2^27 arguments: a uniform grid over [1, 4) (both exponent parities) with random low mantissa bits, and a wide-exponent sample.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include -o rsq_refine_accuracy rsq_refine_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "sctl_amd/device/ukernels.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace sctl_amd;

__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y3, double* y4, double* y5, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RsqConst<double> K;
  y0[i] = rsqrt_masked<0, false>(x[i], K);
  y1[i] = 0.5 * rsqrt_newton2<false>(x[i], K);
  y2[i] = rsqrt_masked<2, false>(x[i], K);
  y3[i] = rsqrt_cubic83<false>(x[i], K);
  y4[i] = rsqrt3_cubic<false>(x[i], K);
  y5[i] = rsqrt5_cubic<false>(x[i], K);
}

int main() {
  CHECK(hipSetDevice(0));
  const int n = 1 << 22, rounds = 32;
  std::vector<double> x(n), h0(n), h1(n), h2(n), h3(n), h4(n), h5(n);
  double *dx, *d0, *d1, *d2, *d3, *d4, *d5;
  CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8)); CHECK(hipMalloc(&d3, n * 8)); CHECK(hipMalloc(&d4, n * 8)); CHECK(hipMalloc(&d5, n * 8));
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(0.0, 1.0), E(-600, 600);
  double m45[2] = {0, 0}, s45[2] = {0, 0}, b45[2] = {0, 0};   // the direct forms (8/15) / r^3 and (8/35) / r^5
  double m3 = 0, s3 = 0, b3 = 0, lo3 = 0, hi3 = 0, c3_lo[3] = {0, 0, 0}, c3_hi[3] = {0, 0, 0}, c3_sq[3] = {0, 0, 0};
  double m0 = 0, m1 = 0, m2 = 0, s0 = 0, s1 = 0, s2 = 0, lo1 = 0, hi1 = 0, bias1 = 0, lo0 = 0, hi0 = 0, bias0 = 0;
  double lo0p[2] = {0, 0}, hi0p[2] = {0, 0};   // seed error by exponent parity of the argument
  double c_lo[3] = {0, 0, 0}, c_hi[3] = {0, 0, 0}, c_sum[3] = {0, 0, 0}, c_sq[3] = {0, 0, 0};   // MODE 1 with its mean folded into the scale, p = 1, 3, 5
  long long cnt = 0, cnt45 = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < n; i++) {
      const double cell = 3.0 / ((double)n * (rounds - 4));
      if (r < rounds - 4) x[i] = 1.0 + ((double)r * n + i + U(g)) * cell;                 // grid over [1, 4), random position inside the cell
      else x[i] = (1.0 + 3.0 * U(g)) * std::pow(2.0, std::floor(E(g)));                    // wide exponent range
    }
    CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, d4, d5, n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), d2, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h3.data(), d3, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h4.data(), d4, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h5.data(), d5, n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
      const long double ex = 1.0L / sqrtl((long double)x[i]);
      const double e0 = (double)((h0[i] - ex) / ex), e1 = (double)((h1[i] - ex) / ex), e2 = (double)((h2[i] - ex) / ex);
      {   // the cubic step as the kernels use it: ((8/3)/r)^p over cubic83_factor(p)
        const long double v = (long double)h3[i] / ex;   // = A (1 + err)
        const double e3 = (double)(v / (long double)cubic83_factor(1) - 1.0L);
        m3 = fmax(m3, fabs(e3)); s3 += e3 * e3; b3 += e3; lo3 = fmin(lo3, e3); hi3 = fmax(hi3, e3);
        for (int q = 0; q < 3; q++) {
          const int p = 2 * q + 1;
          const double ec = (double)(powl(v, p) / (long double)cubic83_factor(p) - 1.0L);
          c3_lo[q] = fmin(c3_lo[q], ec); c3_hi[q] = fmax(c3_hi[q], ec); c3_sq[q] += ec * ec;
        }
      }
      {
        const double e4 = (double)((long double)h4[i] / (ex * ex * ex) / (long double)kCubic3A - 1.0L), e5 = (double)((long double)h5[i] / (ex * ex * ex * ex * ex) / (long double)kCubic5A - 1.0L);
        const long double big = ex * ex * ex * ex * ex;   // the wide-exponent sample reaches r^-5 beyond the range of a double: those are left out
        if (big < 1e300L && big > 1e-300L) {
          m45[0] = fmax(m45[0], fabs(e4)); s45[0] += e4 * e4; b45[0] += e4;
          m45[1] = fmax(m45[1], fabs(e5)); s45[1] += e5 * e5; b45[1] += e5; cnt45++;
        }
      }
      m0 = fmax(m0, fabs(e0)); m1 = fmax(m1, fabs(e1)); m2 = fmax(m2, fabs(e2));
      s0 += e0 * e0; s1 += e1 * e1; s2 += e2 * e2; bias1 += e1; bias0 += e0;
      lo0 = fmin(lo0, e0); hi0 = fmax(hi0, e0);
      { int ex; (void)frexp(x[i], &ex); const int par = ex & 1; lo0p[par] = fmin(lo0p[par], e0); hi0p[par] = fmax(hi0p[par], e0); }
      lo1 = fmin(lo1, e1); hi1 = fmax(hi1, e1);
      for (int q = 0; q < 3; q++) {   // what a kernel with (2/r)^p terms accumulates, over acc_factor = 2^p (1 + p mean): relative error of the pair's value
        const int p = 2 * q + 1;
        const double ec = (double)(powl(1.0L + (long double)e1, p) * (long double)(1 << p) / (long double)newton2_factor(p) - 1.0L);
        c_lo[q] = fmin(c_lo[q], ec); c_hi[q] = fmax(c_hi[q], ec); c_sum[q] += ec; c_sq[q] += ec * ec;
      }
      cnt++;
    }
  }
  printf("%lld arguments\n", cnt);
  printf("MODE 0 seed  (v_rsq_f64):        max rel err %.3e (2^%.2f), rms %.3e, mean %.3e, range [%.3e, %.3e]; by exponent parity [%.3e, %.3e] / [%.3e, %.3e]\n", m0, log2(m0),
         sqrt(s0 / cnt), bias0 / cnt, lo0, hi0, lo0p[0], hi0p[0], lo0p[1], hi0p[1]);
  printf("MODE 1 Newton (unnormalised):    max rel err %.3e (2^%.2f), rms %.3e, mean %.3e, range [%.3e, %.3e]; 3/2 d_max^2 = %.3e\n", m1, log2(m1), sqrt(s1 / cnt),
         bias1 / cnt, lo1, hi1, 1.5 * m0 * m0);
  for (int q = 0; q < 3; q++)
    printf("MODE 1, (2/r)^%d over acc_factor = 2^%d (1 + %d x %.3e):  mean %.3e, rms %.3e, range [%.3e, %.3e]\n", 2 * q + 1, 2 * q + 1, 2 * q + 1, kNewton2MeanErr,
           c_sum[q] / cnt, sqrt(c_sq[q] / cnt), c_lo[q], c_hi[q]);
  printf("MODE 2 cubic, 4 instructions:    max rel err %.3e (2^%.2f = %.2f ulp), rms %.3e, mean %.3e, range [%.3e, %.3e]\n", m3, log2(m3), m3 / 1.1102230246251565e-16,
         sqrt(s3 / cnt), b3 / cnt, lo3, hi3);
  for (int q = 0; q < 3; q++)
    printf("MODE 2 cubic, ((8/3)/r)^%d over cubic83_factor(%d):  rms %.3e, range [%.3e, %.3e]\n", 2 * q + 1, 2 * q + 1, sqrt(c3_sq[q] / cnt), c3_lo[q], c3_hi[q]);
  printf("MODE 2 direct (8/15) / r^3, 5 instructions:  max rel err %.3e (%.2f ulp), rms %.3e, mean %.3e\n", m45[0], m45[0] / 1.1102230246251565e-16, sqrt(s45[0] / cnt45), b45[0] / cnt45);
  printf("MODE 2 direct (8/35) / r^5, 6 instructions:  max rel err %.3e (%.2f ulp), rms %.3e, mean %.3e\n", m45[1], m45[1] / 1.1102230246251565e-16, sqrt(s45[1] / cnt45), b45[1] / cnt45);
  printf("MODE 2 Halley:                   max rel err %.3e (2^%.2f = %.2f ulp), rms %.3e\n", m2, log2(m2), m2 / 1.1102230246251565e-16, sqrt(s2 / cnt));
  return 0;
}
