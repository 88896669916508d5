// Decision microbenchmark for the fp32 tile-centred far loop: can the r^2 of the far pairs come from fp32 MFMA (K = 4 contraction
// [x_t, 1] . [-2 x_s, |x_s|^2] + |x_t|^2) while the VALU does v_rsq_f32 and the packed accumulation?  It pays only if the matrix
// pipe runs BESIDE the fp32 VALU (for fp64 it does not: mfma_mix.hip).  Two-role test: a workgroup of 8 waves = 2 per SIMD; the
// first four waves (one per SIMD) issue only v_mfma_f32_16x16x4_f32, the other four only {2 v_rsq_f32 + v_pk_fma_f32}.  Run each role
// alone and both together: together = max(...) means the pipes overlap, together = sum means they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// roles: bit 0 = MFMA waves active, bit 1 = VALU waves active; VARIANT 0: VALU role = rsq + pk_fma, 1: pk_fma only, 2: rsq only
template <int VARIANT> __global__ void __launch_bounds__(512) k(float* out, int iters, int roles) {
  const int wave = threadIdx.x / 64;
  const bool mfma_role = wave < 4;
  float s = 0;
  if (mfma_role) {
    if (roles & 1) {
      f4 d0 = {1, 2, 3, 4}, d1 = d0, d2 = d0, d3 = d0;
      const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
      for (int it = 0; it < iters; it++) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                     "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
      }
      s = d0.x + d1.y + d2.z + d3.w;
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

template <int VARIANT> void run(const char* what) {
  float* out;
  const int nblk = 256, threads = 512, iters = 40000;
  CHECK(hipMalloc(&out, sizeof(float) * nblk * threads));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms[4] = {0, 0, 0, 0};
  for (int roles = 1; roles <= 3; roles++) {
    hipLaunchKernelGGL(k<VARIANT>, dim3(nblk), dim3(threads), 0, 0, out, 200, roles); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<VARIANT>, dim3(nblk), dim3(threads), 0, 0, out, iters, roles);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms[roles], e0, e1));
  }
  printf("%-44s MFMA waves alone %7.2f ms | VALU waves alone %7.2f ms | both %7.2f ms  (sum %.2f, max %.2f)\n", what, ms[1], ms[2], ms[3], ms[1] + ms[2],
         ms[1] > ms[2] ? ms[1] : ms[2]);
  CHECK(hipFree(out));
}

int main() {
  CHECK(hipSetDevice(0));
  printf("per iteration: MFMA wave = 4 x v_mfma_f32_16x16x4_f32 (r^2 of 1024 pairs); VALU wave = 8 x {...} (the rest of 1024 pairs); one wave of each per SIMD\n");
  run<0>("VALU role: 2 v_rsq_f32 + v_pk_fma_f32");
  run<1>("VALU role: v_pk_fma_f32 only");
  run<2>("VALU role: 2 v_rsq_f32 only");
  return 0;
}
