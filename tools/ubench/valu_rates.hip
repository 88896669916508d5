// Micro-benchmark: per-instruction issue cost of the gfx950 instructions the
// direct-sum kernels are built from (fp64 VALU, fp64 transcendental, converts,
// fp64 MFMA, LDS broadcast reads).  Standalone tool, not part of the product
// path: its numbers justify the instruction budget in DESIGN.md.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
// Output: one line per (instruction, waves/SIMD): shader cycles per wave-instruction
//         as seen by one SIMD (= s_memtime delta * waves_per_simd / instr_per_wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- kernels: each runs `iters` iterations of 4 x 8 independent instructions ----
#define KERNEL_HEAD(name) \
  __global__ void __launch_bounds__(1024) name(double* out, int iters, long long* cyc)
#define PROLOGUE \
  double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7; \
  double b = 1.0000001, c = 1e-9; (void)b; (void)c; \
  float f0 = 1.0f + threadIdx.x * 1e-3f, f1 = f0 + .1f, f2 = f0 + .2f, f3 = f0 + .3f, f4 = f0 + .4f, f5 = f0 + .5f, f6 = f0 + .6f, f7 = f0 + .7f; \
  float fb = 1.0000001f, fc = 1e-9f; (void)fb; (void)fc; \
  long long t0 = __builtin_amdgcn_s_memtime();
#define EPILOGUE \
  long long t1 = __builtin_amdgcn_s_memtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7); \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
#define LOOP4(BODY) for (int i = 0; i < iters; i += 4) { BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY }

#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL_HEAD(k_fma_f64) { PROLOGUE LOOP4(R8(FMA64)) EPILOGUE }
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
KERNEL_HEAD(k_add_f64) { PROLOGUE LOOP4(R8(ADD64)) EPILOGUE }
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL_HEAD(k_mul_f64) { PROLOGUE LOOP4(R8(MUL64)) EPILOGUE }
#define RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
KERNEL_HEAD(k_rsq_f64) { PROLOGUE LOOP4(R8(RSQ64)) EPILOGUE }
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
KERNEL_HEAD(k_rcp_f64) { PROLOGUE LOOP4(R8(RCP64)) EPILOGUE }
#define SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##i));
KERNEL_HEAD(k_sqrt_f64) { PROLOGUE LOOP4(R8(SQRT64)) EPILOGUE }
#define CVT3264(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##i) : "v"(a##i));
KERNEL_HEAD(k_cvt_f32_f64) { PROLOGUE LOOP4(R8(CVT3264)) EPILOGUE }
#define CVT6432(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##i) : "v"(f##i));
KERNEL_HEAD(k_cvt_f64_f32) { PROLOGUE LOOP4(R8(CVT6432)) EPILOGUE }
#define RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(f##i));
KERNEL_HEAD(k_rsq_f32) { PROLOGUE LOOP4(R8(RSQ32)) EPILOGUE }
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##i) : "v"(fb), "v"(fc));
KERNEL_HEAD(k_fma_f32) { PROLOGUE LOOP4(R8(FMA32)) EPILOGUE }
#define EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f##i));
KERNEL_HEAD(k_exp_f32) { PROLOGUE LOOP4(R8(EXP32)) EPILOGUE }
#define SIN32(i) asm volatile("v_sin_f32 %0, %0" : "+v"(f##i));
KERNEL_HEAD(k_sin_f32) { PROLOGUE LOOP4(R8(SIN32)) EPILOGUE }
// compare + 2x cndmask (fp64 select)
#define CMPSEL64(i) asm volatile("v_cmp_lt_f64 vcc, %2, %1\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(f##i) : "v"(a##i), "v"(c) : "vcc");
KERNEL_HEAD(k_cmp_cnd_f64) { PROLOGUE LOOP4(R8(CMPSEL64)) EPILOGUE }
#define CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %1, %0" : : "v"(a##i), "v"(c) : "vcc");
KERNEL_HEAD(k_cmp_f64) { PROLOGUE LOOP4(R8(CMP64)) EPILOGUE }
#define CND32(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(f##i) : "s"(msk));
KERNEL_HEAD(k_cndmask_b32) { PROLOGUE unsigned long long msk = 0xffffffffffffffffull - (unsigned)iters; LOOP4(R8(CND32)) EPILOGUE }
// fp64 FMA with an SGPR operand (source data through the scalar path)
#define FMA64S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "s"(sb), "v"(c));
KERNEL_HEAD(k_fma_f64_sgpr) { PROLOGUE double sb = iters * 1e-12 + 1.0; \
  LOOP4(R8(FMA64S)) EPILOGUE }
// integer/bit op at 32-bit rate
#define AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f##i) : "v"(fb));
KERNEL_HEAD(k_and_b32) { PROLOGUE LOOP4(R8(AND32)) EPILOGUE }
#define PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL_HEAD(k_pk_fma_f32) { PROLOGUE LOOP4(R8(PKFMA32)) EPILOGUE }
#define LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, 0" : "+v"(a##i));
KERNEL_HEAD(k_ldexp_f64) { PROLOGUE LOOP4(R8(LDEXP64)) EPILOGUE }

// fp64 MFMA 16x16x4: 4 independent accumulators, 8 MFMAs per body
KERNEL_HEAD(k_mfma_f64_16x16x4) {
  PROLOGUE
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, a5, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a6, a7, c3, 0, 0, 0);
    }
  }
  a0 += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c2[1] + c3[2];
  EPILOGUE
}
// 32 MFMA + 32*NV fp64 FMAs per iteration in ONE wave: does VALU issue beside the fp64 MFMA?
template <int NV> __device__ __forceinline__ void valu_fill(double& a4, double& a5, double& a6, double& a7, double b, double c) {
#pragma unroll
  for (int v = 0; v < NV; v++) {
    if ((v & 3) == 0) FMA64(4) else if ((v & 3) == 1) FMA64(5) else if ((v & 3) == 2) FMA64(6) else FMA64(7)
  }
}
template <int NV> __global__ void __launch_bounds__(1024) k_mfma_plus_valu(double* out, int iters, long long* cyc) {
  PROLOGUE
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, c0, 0, 0, 0);
      valu_fill<NV>(a4, a5, a6, a7, b, c);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, c1, 0, 0, 0);
      valu_fill<NV>(a4, a5, a6, a7, b, c);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a3, c2, 0, 0, 0);
      valu_fill<NV>(a4, a5, a6, a7, b, c);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a1, c3, 0, 0, 0);
      valu_fill<NV>(a4, a5, a6, a7, b, c);
    }
  }
  a0 += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c2[1] + c3[2];
  EPILOGUE
}
// even waves run MFMA only, odd waves run fp64 FMA only (same instruction count per wave)
KERNEL_HEAD(k_mfma_waves_vs_valu_waves) {
  PROLOGUE
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const int wave = threadIdx.x >> 6;
  if (wave >= 4) {   // waves 4-7 share SIMDs with waves 0-3
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, a5, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a6, a7, c3, 0, 0, 0);
      }
    }
  } else {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 16; u++) { R8(FMA64) R8(FMA64) R8(FMA64) R8(FMA64) }   // 512 FMAs per iteration
    }
  }
  a0 += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c2[1] + c3[2];
  EPILOGUE
}

// LDS broadcast read: every lane reads the same 16 B (the source-tile access pattern)
__global__ void __launch_bounds__(1024) k_lds_bcast_b128(double* out, int iters, long long* cyc) {
  __shared__ double4_t tile[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tile[i] = double4_t{1.0 * i, 2.0, 3.0, 4.0};
  __syncthreads();
  PROLOGUE
  typedef float float4_t __attribute__((ext_vector_type(4)));
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)tile;
  float4_t v0, v1, v2, v3, v4, v5, v6, v7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned ad = base + (((i * 4 + u) * 8) & 1023) * 32;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:32\n ds_read_b128 %2, %8 offset:64\n ds_read_b128 %3, %8 offset:96\n"
                   "ds_read_b128 %4, %8 offset:128\n ds_read_b128 %5, %8 offset:160\n ds_read_b128 %6, %8 offset:192\n ds_read_b128 %7, %8 offset:224\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(ad));
      f0 += v0[0] + v1[1] + v2[2] + v3[3] + v4[0] + v5[1] + v6[2] + v7[3];
    }
  }
  EPILOGUE
}

struct Test { const char* name; void (*fn)(double*, int, long long*); int instr_per_iter; };

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  printf("# device %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int iters = 2000;
  std::vector<Test> tests = {
    {"v_fma_f64", k_fma_f64, 32}, {"v_add_f64", k_add_f64, 32}, {"v_mul_f64", k_mul_f64, 32},
    {"v_fma_f64(sgpr)", k_fma_f64_sgpr, 32}, {"v_ldexp_f64", k_ldexp_f64, 32},
    {"v_rsq_f64", k_rsq_f64, 32}, {"v_rcp_f64", k_rcp_f64, 32}, {"v_sqrt_f64", k_sqrt_f64, 32},
    {"v_cvt_f32_f64", k_cvt_f32_f64, 32}, {"v_cvt_f64_f32", k_cvt_f64_f32, 32},
    {"v_rsq_f32", k_rsq_f32, 32}, {"v_fma_f32", k_fma_f32, 32}, {"v_pk_fma_f32", k_pk_fma_f32, 32},
    {"v_exp_f32", k_exp_f32, 32}, {"v_sin_f32", k_sin_f32, 32}, {"v_and_b32", k_and_b32, 32},
    {"v_cmp_lt_f64", k_cmp_f64, 32}, {"v_cndmask_b32", k_cndmask_b32, 32}, {"cmp_f64+cndmask", k_cmp_cnd_f64, 32},
    {"mfma_f64_16x16x4", k_mfma_f64_16x16x4, 32},
    {"mfma_f64+0valu(1 wave stream, per mfma)", k_mfma_plus_valu<0>, 32},
    {"mfma_f64+4valu(per mfma)", k_mfma_plus_valu<4>, 32},
    {"mfma_f64+8valu(per mfma)", k_mfma_plus_valu<8>, 32},
    {"mfma_f64+12valu(per mfma)", k_mfma_plus_valu<12>, 32},
    {"mfma_f64+16valu(per mfma)", k_mfma_plus_valu<16>, 32},
    {"mfma waves 4-7 (32/it) beside valu waves 0-3 (512/it): per it", k_mfma_waves_vs_valu_waves, 1},
    {"ds_read_b128 bcast", k_lds_bcast_b128, 32},
  };
  const int nblk = prop.multiProcessorCount;   // one block per CU
  for (auto& t : tests) {
    for (int wps : {1, 2, 4}) {                 // waves per SIMD
      const bool two_role = (t.instr_per_iter == 1);
      if (two_role && wps != 2) continue; if (hipGetLastError() != hipSuccess) {}
      const int threads = 256 * wps;
      double* out; long long* cyc;
      CHECK(hipMalloc(&out, sizeof(double) * nblk * threads));
      CHECK(hipMalloc(&cyc, sizeof(long long) * nblk * threads / 64));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(t.fn, dim3(nblk), dim3(threads), 0, 0, out, 52, cyc);   // warm-up
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.fn, dim3(nblk), dim3(threads), 0, 0, out, iters, cyc);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(nblk * threads / 64);
      CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      if (two_role) {
        std::vector<long long> hv, hm;
        for (size_t w = 0; w < h.size(); w++) ((w % 8) < 4 ? hv : hm).push_back(h[w]);
        std::sort(hv.begin(), hv.end()); std::sort(hm.begin(), hm.end());
        printf("%-64s VALU waves: %8.2f cycles/iter (512 fma => %.2f/fma)   MFMA waves: %8.2f cycles/iter (32 mfma => %.2f/mfma)  kernel_ms=%8.3f\n",
               t.name, (double)hv[hv.size() / 2] / iters, (double)hv[hv.size() / 2] / iters / 512, (double)hm[hm.size() / 2] / iters,
               (double)hm[hm.size() / 2] / iters / 32, ms);
      } else {
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      const double n_instr = (double)iters * t.instr_per_iter;
      printf("%-64s waves/SIMD=%d  cycles/wave-instr(one wave)=%8.2f  cycles/instr/SIMD=%7.2f  kernel_ms=%8.3f  eff_clock_GHz=%5.2f\n",
             t.name, wps, med / n_instr, med / n_instr / wps, ms, med / (ms * 1e6));
      }
      CHECK(hipFree(out)); CHECK(hipFree(cyc));
    }
  }
  return 0;
}
