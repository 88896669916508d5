// Exploration tool (round 4): the 1/r seed of the fp64 tile-centred Laplace far loop taken from the bf16 MATRIX cores.
//   BASE   the shipped loop body: 4 (centred distance) + v_rsq_f64 (16 cycles) + 4 (cubic step) + 1 (accumulate), four targets per lane (centered_kernel.hpp)
//   MFMA   r2 of 32 sources x 32 targets in fp32 from two v_mfma_f32_32x32x16_bf16 on split-bf16 operands (centered_mfma_kernel.hpp) -> v_rsq_f32 ->
//          v_cvt_f64_f32 = a ~20-bit seed; the exact fp64 r2, the four-instruction cubic step and the accumulation stay on the fp64 vector pipe.  A lane then
//          holds one target column per block of 32 targets and 16 of a block's 32 source rows (the MFMA's output layout): the two half-waves read different
//          records from LDS, and each staged source also gets a 64-byte row of bf16 pieces.
//   MFMA-A the same with the seed's bits shifted into a double's high word by one v_alignbit_b32 instead of the conversion (coordinates scaled by 2^-128).
// Both time the same problem with the same staging (no far / near split: every source is far); what differs is the loop body and, for MFMA, the row staging.
// Costed beforehand from measured rates: 1.7 (MFMA issue per wave-pair) + 8 (v_rsq_f32) + 4 (conversion) = 13.7 against 16 cycles for v_rsq_f64, i.e. at most
// 2.3 of ~54.8 cycles per wave-pair.  Taken into the product only if the loop gains >= 3 %.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include -I sctl_amd/csrc tools/ubench/mfma_seed.hip -o tools/ubench/mfma_seed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "centered_mfma_kernel.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace sctl_amd;
typedef double d2 __attribute__((ext_vector_type(2)));

// (8/3)/r from a seed y0 (any accuracy >= ~18 bits), r2 exact: rsqrt_cubic83 of ukernels.hpp without the seed
__device__ __forceinline__ double cubic_from_seed(double r2, double y0, const RsqConst<double>& K) {
  const double z = __builtin_fma(r2, y0 * y0, -K.c53);
  return y0 * __builtin_fma(z, z, K.k209);
}

// ---- BASE: the shipped far loop (T targets per lane, U sources per trip) ------------------------------------------------------------------
template <int T, int U> __global__ void __launch_bounds__(64) lap_base(const double* __restrict__ xt, const double* __restrict__ xs, const double* __restrict__ fs,
                                                                       double* __restrict__ out, int Ns) {
  constexpr int TILE = 64;
  __shared__ d2 tile[TILE * 2];
  __shared__ double dens[TILE];
  const int tid = threadIdx.x;
  const RsqConst<double> K;
  double m2x[T][3], tt[T], acc[T];
#pragma unroll
  for (int j = 0; j < T; j++) {
    const long t = (long)blockIdx.x * (64 * T) + j * 64 + tid;
    double x[3];
    for (int k = 0; k < 3; k++) x[k] = xt[t * 3 + k] - 0.5;
    tt[j] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    for (int k = 0; k < 3; k++) m2x[j][k] = -2 * x[k];
    acc[j] = 0;
  }
  double px = xs[tid * 3] + 1.0, py = xs[tid * 3 + 1] + 1.0, pz = xs[tid * 3 + 2] + 1.0, pf = fs[tid];   // sources in [1,2)^3: far from the targets' cube about 0
  for (int s0 = 0; s0 < Ns; s0 += TILE) {
    __syncthreads();
    tile[tid * 2] = d2{px, py};
    tile[tid * 2 + 1] = d2{pz, px * px + py * py + pz * pz};
    dens[tid] = pf;
    if (s0 + TILE < Ns) { const long s = s0 + TILE + tid; px = xs[s * 3] + 1.0; py = xs[s * 3 + 1] + 1.0; pz = xs[s * 3 + 2] + 1.0; pf = fs[s]; }
    __syncthreads();
#pragma unroll U
    for (int s = 0; s < TILE; s++) {
      const d2 a = tile[s * 2], b = tile[s * 2 + 1];
      const double f = dens[s];
#pragma unroll
      for (int j = 0; j < T; j++) {
        const double r2 = __builtin_fma(m2x[j][0], a[0], __builtin_fma(m2x[j][1], a[1], __builtin_fma(m2x[j][2], b[0], tt[j] + b[1])));
        acc[j] = __builtin_fma(f, cubic_from_seed(r2, __builtin_amdgcn_rsq(r2), K), acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < T; j++) out[(long)blockIdx.x * (64 * T) + j * 64 + tid] = acc[j];
}

// ---- MFMA: the seed from the matrix cores --------------------------------------------------------------------------------------------------
// CB column blocks of 32 targets per wave; lane (m = lane % 32, h = lane / 32) owns column m of every block and, per block of 32 source rows, the rows
// 8 k + 4 h + {0..3}.  ALIGN: seed bits -> high word of a double by v_alignbit_b32 (everything scaled by 2^-128) instead of v_cvt_f64_f32.
template <int CB, bool ALIGN, int WAVES> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
lap_mfma(const double* __restrict__ xt, const double* __restrict__ xs, const double* __restrict__ fs, double* __restrict__ out, int Ns) {
  constexpr int TILE = 64, RW = 5;
  __shared__ u32x4 farA[TILE * RW];
  __shared__ d2 rec[TILE * 2];
  __shared__ double dens[TILE];
  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  const RsqConst<double> K;
  const double cs = ALIGN ? 0x1p-128 : 1.0;
  unsigned cout = 2u;
  asm volatile("" : "+s"(cout));
  double m2x[CB][3], tt[CB], acc[CB];
  u32x4 Bop[CB][2];
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    const long t = (long)blockIdx.x * (32 * CB) + cb * 32 + m;
    double x[3];
    for (int k = 0; k < 3; k++) x[k] = xt[t * 3 + k] - 0.5;
    const float xf[3] = {(float)x[0], (float)x[1], (float)x[2]};
    const float ttf = xf[0] * xf[0] + xf[1] * xf[1] + xf[2] * xf[2];
    const u32x4 w[4] = {b_word(-2.0f * xf[0]), b_word(-2.0f * xf[1]), b_word(-2.0f * xf[2]), b_tail(ttf)};
#pragma unroll
    for (int step = 0; step < 2; step++)
#pragma unroll
      for (int i = 0; i < 4; i++) Bop[cb][step][i] = h ? w[2 * step + 1][i] : w[2 * step][i];
    for (int k = 0; k < 3; k++) x[k] *= cs;
    tt[cb] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    for (int k = 0; k < 3; k++) m2x[cb][k] = -2 * x[k];
    acc[cb] = 0;
  }
  auto mfma = [](u32x4 A, u32x4 B, f32x16 C) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0); };
  const f32x16 zero = {};
  double px = xs[lane * 3] + 1.0, py = xs[lane * 3 + 1] + 1.0, pz = xs[lane * 3 + 2] + 1.0, pf = fs[lane];
  const d2* const rec_h = rec + h * 8;          // this half-wave's rows 4 h + ... of every group of eight
  const double* const dens_h = dens + h * 4;
  for (int s0 = 0; s0 < Ns; s0 += TILE) {
    __syncthreads();
    {
      const float xf = (float)px, yf = (float)py, zf = (float)pz;
      u32x4* row = farA + lane * RW;
      row[0] = a_word(xf);
      row[1] = a_word(yf);
      row[2] = a_word(zf);
      row[3] = a_tail(xf * xf + yf * yf + zf * zf, true);
      const double x = px * cs, y = py * cs, z = pz * cs;
      rec[lane * 2] = d2{x, y};
      rec[lane * 2 + 1] = d2{z, x * x + y * y + z * z};
      dens[lane] = pf;
    }
    if (s0 + TILE < Ns) { const long s = s0 + TILE + lane; px = xs[s * 3] + 1.0; py = xs[s * 3 + 1] + 1.0; pz = xs[s * 3 + 2] + 1.0; pf = fs[s]; }
    __syncthreads();
#pragma unroll 1
    for (int r0 = 0; r0 < TILE; r0 += 32) {
      const u32x4* row = farA + (r0 + m) * RW + h;
      const u32x4 A0 = row[0], A1 = row[2];
      f32x16 r2f[CB];
#pragma unroll
      for (int cb = 0; cb < CB; cb++) r2f[cb] = mfma(A1, Bop[cb][1], mfma(A0, Bop[cb][0], zero));
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int s = r0 + 8 * k + i;           // + 4 h through rec_h / dens_h
          const d2 a = rec_h[s * 2], b = rec_h[s * 2 + 1];
          const double f = dens_h[s];
#pragma unroll
          for (int cb = 0; cb < CB; cb++) {
            const double r2 = __builtin_fma(m2x[cb][0], a[0], __builtin_fma(m2x[cb][1], a[1], __builtin_fma(m2x[cb][2], b[0], tt[cb] + b[1])));
            const float yf = __builtin_amdgcn_rsqf(r2f[cb][4 * k + i]);
            double y0;
            if (ALIGN) {
              unsigned lo;
              asm("" : "=v"(lo) : "v"(yf));     // any bits: they sit below the seed's accuracy
              y0 = __hiloint2double((int)__builtin_amdgcn_alignbit(cout, __float_as_uint(yf), 3), (int)lo);   // = yf 2^128 (20 mantissa bits)
            } else {
              y0 = (double)yf;
            }
            acc[cb] = __builtin_fma(f, cubic_from_seed(r2, y0, K), acc[cb]);
          }
        }
      }
      asm volatile("" ::"v"(A0), "v"(A1));
    }
  }
#pragma unroll
  for (int cb = 0; cb < CB; cb++) asm volatile("" ::"v"(Bop[cb][0]), "v"(Bop[cb][1]));
#pragma unroll
  for (int cb = 0; cb < CB; cb++) {
    acc[cb] += __shfl_xor(acc[cb], 32);
    if (h == 0) out[(long)blockIdx.x * (32 * CB) + cb * 32 + m] = acc[cb] * cs;
  }
}

struct Var { const char* name; void (*fn)(const double*, const double*, const double*, double*, int); int per_wave; };

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1 << 20;
  std::vector<double> h(N * 3); for (auto& v : h) v = drand48();
  double *xt, *xs, *f, *out;
  CHECK(hipMalloc(&xt, N * 24)); CHECK(hipMalloc(&xs, N * 24)); CHECK(hipMalloc(&f, N * 8)); CHECK(hipMalloc(&out, N * 8));
  CHECK(hipMemcpy(xt, h.data(), N * 24, hipMemcpyHostToDevice));
  for (auto& v : h) v = drand48();
  CHECK(hipMemcpy(xs, h.data(), N * 24, hipMemcpyHostToDevice));
  for (auto& v : h) v = drand48() - 0.5;
  CHECK(hipMemcpy(f, h.data(), N * 8, hipMemcpyHostToDevice));
  std::vector<Var> vars = {
      {"BASE  T=4 U=4 (shipped body)", lap_base<4, 4>, 256},
      {"BASE  T=2 U=4", lap_base<2, 4>, 128},
      {"MFMA  CB=4 cvt   2 waves", lap_mfma<4, false, 2>, 128},
      {"MFMA  CB=4 cvt   3 waves", lap_mfma<4, false, 3>, 128},
      {"MFMA  CB=4 align 2 waves", lap_mfma<4, true, 2>, 128},
      {"MFMA  CB=2 cvt   4 waves", lap_mfma<2, false, 4>, 64},
      {"MFMA  CB=2 cvt   3 waves", lap_mfma<2, false, 3>, 64},
      {"BASE  T=4 U=4 (again)", lap_base<4, 4>, 256},
  };
  std::vector<double> ref(N), got(N);
  for (size_t i = 0; i < vars.size(); i++) {
    auto& v = vars[i];
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(N / v.per_wave), dim3(64), 0, 0, xt, xs, f, out, N);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(got.data(), out, N * 8, hipMemcpyDeviceToHost));
    if (i == 0) ref = got;
    double d = 0, n = 0, mx = 0; for (int k = 0; k < N; k++) { d += (got[k] - ref[k]) * (got[k] - ref[k]); n += ref[k] * ref[k]; mx = fmax(mx, fabs(got[k] - ref[k]) / fabs(ref[k])); }
    const double pps = (double)N * N / (best * 1e-3);
    printf("%-30s %8.2f ms  %.3e pairs/s  %5.1f%% of 78.6 TF   rel-L2 vs first %.2e  max rel %.2e\n", v.name, best, pps, pps * 11 / 78.6e12 * 100, sqrt(d / n), mx);
    fflush(stdout);
  }
  return 0;
}
