// Decision microbenchmark for an fp32 far loop whose r^2 comes from the bf16 matrix cores: coordinates split into three bf16 pieces each, the 18 cross
// products that matter + |x_t'|^2 + |x_s'|^2 as a K = 24 (padded 32) contraction = 2 x v_mfma_f32_32x32x16_bf16 per 32 sources x 32 targets, fp32
// accumulate.  fp32-input MFMA and the fp32 VALU exclude each other on gfx950 (f32_mfma_mix.hip); MI355X_MICROARCH.md says a bf16 MFMA holds the SIMD's
// vector issue for only 8 of its 32 cycles.  Two tests:
//   (1) two roles, one wave of each per SIMD: MFMA-only waves beside {v_rsq_f32, v_pk_fma_f32} waves — alone and together;
//   (2) ONE wave doing both, as the kernel would: per 1024 pairs 2 MFMAs (this trip's r^2 into one accumulator set) while the VALU works on the
//       previous trip's set: 16 v_rsq_f32 + 8 v_pk_fma_f32 per lane — against the same VALU work without the MFMAs and against the shipped loop's
//       mix (per 16 pairs: 8 + 24 v_pk_* for the distance, 16 v_rsq_f32, 8 v_pk_fma_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16v mfma(bf8 a, bf8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// (1) two roles
template <int VARIANT> __global__ void __launch_bounds__(512) two_roles(float* out, int iters, int roles) {
  const int wave = threadIdx.x / 64;
  float s = 0;
  if (wave < 4) {
    if (roles & 1) {
      bf8 a, b;
      for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
      f16v c0 = {}, c1 = {};
      for (int it = 0; it < iters; it++) {
        c0 = mfma(a, b, c0); c1 = mfma(a, b, c1);
        asm volatile("" : "+v"(c0), "+v"(c1));
      }
      s = c0[0] + c1[5];
    }
  } else if (roles & 2) {
    f2 p[8], b2 = {1.0000001f, 0.9999999f}, c2 = {1e-9f, 2e-9f};
    for (int i = 0; i < 8; i++) p[i] = f2{1.0f + threadIdx.x * 1e-6f + i, 2.0f + i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (VARIANT != 1) { asm volatile("v_rsq_f32 %0, %0" : "+v"(p[u].x)); asm volatile("v_rsq_f32 %0, %0" : "+v"(p[u].y)); }
        if (VARIANT != 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(b2), "v"(c2));
      }
    }
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (2) one wave does both.  KIND 0: MFMA r^2 + rsq + accumulate; 1: the same VALU work, no MFMA (r^2 taken as given); 2: the shipped mix (packed VALU distance)
template <int KIND> __global__ void __launch_bounds__(256) one_wave(float* out, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(0.5f + 0.01f * i); }
  f16v cur = {}, nxt = {};
  for (int i = 0; i < 16; i++) cur[i] = 1.0f + i + threadIdx.x * 1e-4f;
  f2 acc[2] = {f2{0, 0}, f2{0, 0}};
  const f2 f = {1.0000001f, 0.9999999f};
  f2 m0 = {0.1f, 0.2f}, m1 = {0.3f, 0.4f}, m2 = {0.5f, 0.6f}, tt = {1.5f, 2.5f};
  asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(tt));
  auto process = [&](const f16v& c) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      f2 r2 = {c[2 * u], c[2 * u + 1]};
      if (KIND == 2) {   // the packed distance of the shipped loop: per two pairs 1 v_pk_add + 3 v_pk_fma
        f2 s = f2{c[2 * u], c[2 * u]}, q = tt + f2{c[2 * u + 1], c[2 * u + 1]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(m2), "v"(s));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(m1), "v"(s));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(m0), "v"(s));
        r2 = q;
      }
      f2 y;
      asm volatile("v_rsq_f32 %0, %1" : "=v"(y.x) : "v"(r2.x));
      asm volatile("v_rsq_f32 %0, %1" : "=v"(y.y) : "v"(r2.y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(f), "v"(y));
    }
  };
  for (int it = 0; it < iters; it += 2) {   // two trips per round: the sets swap roles, no copies
    // (results asked for in VGPRs: left to itself the compiler puts MFMA results in AGPRs, which the VALU cannot read without a v_accvgpr_read each)
    if (KIND == 0) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0\n v_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "=&v"(nxt) : "v"(a), "v"(b)); }
    process(cur);
    if (KIND == 0) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, 0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "=&v"(cur) : "v"(a), "v"(b)); }
    else { for (int i = 0; i < 16; i++) nxt[i] = cur[i] + 1e-3f; }
    process(nxt);
    if (KIND != 0) { for (int i = 0; i < 16; i++) cur[i] = nxt[i] + 1e-3f; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0].x + acc[0].y + acc[1].x + acc[1].y + cur[3];
}

template <int VARIANT> void run_roles(const char* what) {
  float* out;
  const int nblk = 256, threads = 512, iters = 40000;
  CHECK(hipMalloc(&out, sizeof(float) * nblk * threads));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms[4] = {0, 0, 0, 0};
  for (int roles = 1; roles <= 3; roles++) {
    hipLaunchKernelGGL(two_roles<VARIANT>, dim3(nblk), dim3(threads), 0, 0, out, 200, roles); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(two_roles<VARIANT>, dim3(nblk), dim3(threads), 0, 0, out, iters, roles);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms[roles], e0, e1));
  }
  printf("%-44s MFMA waves alone %7.2f ms | VALU waves alone %7.2f ms | both %7.2f ms  (sum %.2f, max %.2f)\n", what, ms[1], ms[2], ms[3], ms[1] + ms[2],
         ms[1] > ms[2] ? ms[1] : ms[2]);
  CHECK(hipFree(out));
}
template <int KIND> float run_one(const char* what, int waves_per_simd) {
  float* out;
  const int nblk = 256 * waves_per_simd, threads = 256, iters = 20000;
  CHECK(hipMalloc(&out, sizeof(float) * nblk * threads));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(one_wave<KIND>, dim3(nblk), dim3(threads), 0, 0, out, 200); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(one_wave<KIND>, dim3(nblk), dim3(threads), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // per iteration a wave does 16 pairs per lane = 16 wave-pairs; cycles per wave-pair per SIMD at 2.4 GHz (nominal; the clock under load is lower)
  printf("%-72s %d wave(s)/SIMD: %7.2f ms  = %.1f nominal cycles per wave-pair per SIMD\n", what, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * waves_per_simd));
  CHECK(hipFree(out));
  return ms;
}

int main() {
  CHECK(hipSetDevice(0));
  printf("(1) per iteration: MFMA wave = 2 x v_mfma_f32_32x32x16_bf16 (r^2 of 1024 pairs, K = 32); VALU wave = 8 x {...} (the rest of 1024 pairs); one wave of each per SIMD\n");
  run_roles<0>("VALU role: 2 v_rsq_f32 + v_pk_fma_f32");
  run_roles<1>("VALU role: v_pk_fma_f32 only");
  run_roles<2>("VALU role: 2 v_rsq_f32 only");
  printf("(2) one wave does both; per iteration 16 pairs per lane\n");
  for (int w = 1; w <= 4; w *= 2) {
    run_one<0>("2 bf16 MFMA (next r^2) + 16 v_rsq_f32 + 8 v_pk_fma_f32", w);
    run_one<1>("16 v_rsq_f32 + 8 v_pk_fma_f32 (r^2 given)", w);
    run_one<2>("shipped mix: 24 v_pk_fma + 8 v_pk_add (distance) + 16 v_rsq_f32 + 8 v_pk_fma", w);
  }
  return 0;
}
