// Does a VALU instruction that overwrites the SOURCE register of the transcendental instruction issued just before it reach the register before the
// transcendental unit has read it?  (Found as wrong, run-to-run different near-field sums in lanes 48-63 of centered_mfma_f32_kernel<true, 4>: hipcc
// had emitted `v_rsq_f32 v34, v36` directly followed by `v_pk_mul_f32 v[36:37], ...`; ROCm 7.2 inserts wait states for the read-after-write case only.)
// Each lane runs   y = rsq(x);  x = <something else>   back to back, many times, and compares y with rsq of the x it should have seen.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/trans_war.hip -o tools/ubench/trans_war && tools/ubench/trans_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// FOLLOW: 0 = v_mov_b32 into the source, 1 = v_pk_mul_f32 into the source pair, 2 = v_mul_f32 into the source, 3 = v_pk_mul_f32 after one s_nop 0,
//         4 = v_mov_b32 after one independent VALU instruction
template <int FOLLOW, int BACKLOG> __global__ void __launch_bounds__(256) kern(const float* in, unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63;
  float x0 = in[blockIdx.x * blockDim.x + threadIdx.x];
  unsigned nbad = 0;
  float other = 3.0f + lane;
  for (int it = 0; it < iters; it++) {
    const float xin = x0 + it;
    const float expect = __builtin_amdgcn_rsqf(xin);
    const f2 o2 = {other, other};
    float y;
    // the source lives in v40 (v[40:41] for the packed follower); a few idle cycles before, so that nothing else is in flight
#define PRE0 "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\ts_nop 7\n\t"
#define RSQ4 "v_rsq_f32 v42, v44\n\tv_rsq_f32 v43, v45\n\tv_rsq_f32 v46, v44\n\tv_rsq_f32 v47, v45\n\t"
    asm volatile("v_mov_b32 v44, %0\n\tv_mov_b32 v45, %0" ::"v"(xin) : "v44", "v45");
#define CL : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
    if (BACKLOG == 0) {
      if (FOLLOW == 0) asm volatile(PRE0 "v_rsq_f32 %0, v40\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 1) asm volatile(PRE0 "v_rsq_f32 %0, v40\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 2) asm volatile(PRE0 "v_rsq_f32 %0, v40\n\tv_mul_f32 v40, %2, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 3) asm volatile(PRE0 "v_rsq_f32 %0, v40\n\ts_nop 0\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 4) asm volatile(PRE0 "v_rsq_f32 %0, v40\n\tv_add_f32 v41, %2, %2\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
    }
    if (BACKLOG == 4) {
      if (FOLLOW == 0) asm volatile(PRE0 RSQ4 "v_rsq_f32 %0, v40\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 1) asm volatile(PRE0 RSQ4 "v_rsq_f32 %0, v40\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 2) asm volatile(PRE0 RSQ4 "v_rsq_f32 %0, v40\n\tv_mul_f32 v40, %2, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 3) asm volatile(PRE0 RSQ4 "v_rsq_f32 %0, v40\n\ts_nop 0\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 4) asm volatile(PRE0 RSQ4 "v_rsq_f32 %0, v40\n\tv_add_f32 v41, %2, %2\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
    }
    if (BACKLOG == 8) {
      if (FOLLOW == 0) asm volatile(PRE0 RSQ4 RSQ4 "v_rsq_f32 %0, v40\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 1) asm volatile(PRE0 RSQ4 RSQ4 "v_rsq_f32 %0, v40\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 2) asm volatile(PRE0 RSQ4 RSQ4 "v_rsq_f32 %0, v40\n\tv_mul_f32 v40, %2, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 3) asm volatile(PRE0 RSQ4 RSQ4 "v_rsq_f32 %0, v40\n\ts_nop 0\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
      if (FOLLOW == 4) asm volatile(PRE0 RSQ4 RSQ4 "v_rsq_f32 %0, v40\n\tv_add_f32 v41, %2, %2\n\tv_mov_b32 v40, %2\n\ts_nop 7" : "=&v"(y) : "v"(xin), "v"(other), "v"(o2)  CL);
    }
    if (y != expect) nbad++;
    other += y * 1e-30f;
  }
  atomicAdd(&bad[lane >> 4], nbad);
}

template <int FOLLOW, int BACKLOG> void run(const char* what, int waves_per_simd) {
  const int nblk = 256 * waves_per_simd, threads = 256, iters = 20000;
  std::vector<float> h((size_t)nblk * threads);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + (float)(i % 977) * 0.37f;
  float* in; unsigned* bad;
  CHECK(hipMalloc(&in, h.size() * sizeof(float))); CHECK(hipMalloc(&bad, 4 * sizeof(unsigned)));
  CHECK(hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice)); CHECK(hipMemset(bad, 0, 4 * sizeof(unsigned)));
  hipLaunchKernelGGL((kern<FOLLOW, BACKLOG>), dim3(nblk), dim3(threads), 0, 0, in, bad, iters);
  CHECK(hipDeviceSynchronize());
  unsigned hb[4]; CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
  const double total = (double)nblk * threads * iters / 4;
  printf("%-62s after %d other v_rsq_f32, %d wave(s)/SIMD: wrong results, lanes 0-15 / 16-31 / 32-47 / 48-63: %.3g / %.3g / %.3g / %.3g of the evaluations\n", what, BACKLOG, waves_per_simd, hb[0] / total,
         hb[1] / total, hb[2] / total, hb[3] / total);
  CHECK(hipFree(in)); CHECK(hipFree(bad));
}

// A second shape, as in the compiled near loop: an EARLIER transcendental, GAP independent VALU instructions (4 cycles each; the transcendental unit needs 8),
// the tested v_rsq_f32, and directly behind it the instruction that overwrites its source (PK: v_pk_mul_f32 on the pair, else v_mov_b32).
template <int GAP, bool PK> __global__ void __launch_bounds__(256) kern2(const float* in, unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63;
  float x0 = in[blockIdx.x * blockDim.x + threadIdx.x];
  unsigned nbad = 0;
  float other = 3.0f + lane;
  for (int it = 0; it < iters; it++) {
    const float xin = x0 + it;
    const float expect = __builtin_amdgcn_rsqf(xin);
    const f2 o2 = {other, other};
    float y;
#define HEAD "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v44, %1\n\ts_nop 7\n\tv_rsq_f32 v42, v44\n\t"
#define FILL "v_add_f32 v46, %2, %2\n\t"
#define TAILPK "v_rsq_f32 %0, v40\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7"
#define TAILMV "v_rsq_f32 %0, v40\n\tv_mov_b32 v40, %2\n\ts_nop 7"
#define OPS : "=&v"(y) : "v"(xin), "v"(other), "v"(o2) : "v40", "v41", "v42", "v44", "v46"
    if (GAP == 0 && PK) asm volatile(HEAD TAILPK OPS);
    if (GAP == 1 && PK) asm volatile(HEAD FILL TAILPK OPS);
    if (GAP == 2 && PK) asm volatile(HEAD FILL FILL TAILPK OPS);
    if (GAP == 3 && PK) asm volatile(HEAD FILL FILL FILL TAILPK OPS);
    if (GAP == 0 && !PK) asm volatile(HEAD TAILMV OPS);
    if (GAP == 1 && !PK) asm volatile(HEAD FILL TAILMV OPS);
    if (GAP == 2 && !PK) asm volatile(HEAD FILL FILL TAILMV OPS);
    if (GAP == 3 && !PK) asm volatile(HEAD FILL FILL FILL TAILMV OPS);
    if (y != expect) nbad++;
    other += y * 1e-30f;
  }
  atomicAdd(&bad[lane >> 4], nbad);
}
template <int GAP, bool PK> void run2(int waves_per_simd) {
  const int nblk = 256 * waves_per_simd, threads = 256, iters = 20000;
  std::vector<float> h((size_t)nblk * threads);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + (float)(i % 977) * 0.37f;
  float* in; unsigned* bad;
  CHECK(hipMalloc(&in, h.size() * sizeof(float))); CHECK(hipMalloc(&bad, 4 * sizeof(unsigned)));
  CHECK(hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice)); CHECK(hipMemset(bad, 0, 4 * sizeof(unsigned)));
  hipLaunchKernelGGL((kern2<GAP, PK>), dim3(nblk), dim3(threads), 0, 0, in, bad, iters);
  CHECK(hipDeviceSynchronize());
  unsigned hb[4]; CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
  const double total = (double)nblk * threads * iters / 4;
  printf("v_rsq_f32 ; %d x v_add_f32 ; v_rsq_f32 y, x ; %-28s %d wave(s)/SIMD: wrong, lanes 0-15 / 16-31 / 32-47 / 48-63: %.3g / %.3g / %.3g / %.3g\n", GAP,
         PK ? "v_pk_mul_f32 {x, x'}, o, o" : "v_mov_b32 x, other", waves_per_simd, hb[0] / total, hb[1] / total, hb[2] / total, hb[3] / total);
  CHECK(hipFree(in)); CHECK(hipFree(bad));
}

// A third shape: the hypothesis that explains the fault of DESIGN.md §4.2a.  Two roles share every SIMD (512-thread workgroups: waves w and w + 4):
//   role B (waves 4..7): bursts of BURST back-to-back independent v_rsq_f32 — what the batched far loop of the matrix-core double-layer kernel issues;
//   role A (waves 0..3): y = rsq(x) DIRECTLY followed by an instruction that overwrites x (PK: v_pk_mul_f32 on the pair, else v_mov_b32), then idle.
// If a transcendental reads its source when it reaches the head of the SIMD's transcendental queue rather than at issue, role A's rsq — waiting behind
// role B's burst — sees the overwritten register.
template <int BURST, bool PK, int GAPNOPS> __global__ void __launch_bounds__(512) kern3(const float* in, unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x0 = in[blockIdx.x * blockDim.x + threadIdx.x];
  unsigned nbad = 0;
  float other = 3.0f + lane;
  if (wave >= 4) {   // role B
    float a0 = x0, a1 = x0 + 1, a2 = x0 + 2, a3 = x0 + 3, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int it = 0; it < iters * 2; it++) {
#pragma unroll
      for (int b = 0; b < BURST; b += 4)
        asm volatile("v_rsq_f32 %0, %4\n\tv_rsq_f32 %1, %5\n\tv_rsq_f32 %2, %6\n\tv_rsq_f32 %3, %7" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      a0 += r0 * 1e-30f;
    }
    if (a0 == 12345.0f) bad[0] = 1;
    return;
  }
  for (int it = 0; it < iters; it++) {
    const float xin = x0 + it;
    const float expect = __builtin_amdgcn_rsqf(xin);
    const f2 o2 = {other, other};
    float y;
#define HEAD3 "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\ts_nop 3\n\tv_rsq_f32 %0, v40\n\t"
#define OPS3 : "=&v"(y) : "v"(xin), "v"(other), "v"(o2) : "v40", "v41"
    if (PK && GAPNOPS == 0) asm volatile(HEAD3 "v_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" OPS3);
    if (PK && GAPNOPS == 1) asm volatile(HEAD3 "s_nop 0\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" OPS3);
    if (PK && GAPNOPS == 16) asm volatile(HEAD3 "s_nop 15\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7" OPS3);
    if (!PK && GAPNOPS == 0) asm volatile(HEAD3 "v_mov_b32 v40, %2\n\ts_nop 7" OPS3);
    if (y != expect) nbad++;
    other += y * 1e-30f;
  }
  atomicAdd(&bad[lane >> 4], nbad);
}
template <int BURST, bool PK, int GAPNOPS> void run3() {
  const int nblk = 256, threads = 512, iters = 20000;
  std::vector<float> h((size_t)nblk * threads);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + (float)(i % 977) * 0.37f;
  float* in; unsigned* bad;
  CHECK(hipMalloc(&in, h.size() * sizeof(float))); CHECK(hipMalloc(&bad, 4 * sizeof(unsigned)));
  CHECK(hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice)); CHECK(hipMemset(bad, 0, 4 * sizeof(unsigned)));
  hipLaunchKernelGGL((kern3<BURST, PK, GAPNOPS>), dim3(nblk), dim3(threads), 0, 0, in, bad, iters);
  CHECK(hipDeviceSynchronize());
  unsigned hb[4]; CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
  const double total = (double)nblk * 256 * iters / 4;
  printf("beside a wave issuing bursts of %2d v_rsq_f32:  v_rsq_f32 y, x ; %s%-26s wrong, lanes 0-15 / 16-31 / 32-47 / 48-63: %.3g / %.3g / %.3g / %.3g\n", BURST,
         GAPNOPS == 0 ? "" : (GAPNOPS == 1 ? "s_nop 0 ; " : "s_nop 15 ; "), PK ? "v_pk_mul_f32 {x, x'}, o, o" : "v_mov_b32 x, other", hb[0] / total, hb[1] / total, hb[2] / total, hb[3] / total);
  CHECK(hipFree(in)); CHECK(hipFree(bad));
}

int main() {
  CHECK(hipSetDevice(0));
  run3<4, true, 0>(); run3<16, true, 0>(); run3<32, true, 0>(); run3<32, false, 0>(); run3<32, true, 1>(); run3<32, true, 16>();
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 0>("v_rsq_f32 y, x ; v_mov_b32 x, other", w);
    run<1, 0>("v_rsq_f32 y, x ; v_pk_mul_f32 {x, x'}, o, o", w);
    run<0, 4>("v_rsq_f32 y, x ; v_mov_b32 x, other", w);
    run<1, 4>("v_rsq_f32 y, x ; v_pk_mul_f32 {x, x'}, o, o", w);
    run<2, 4>("v_rsq_f32 y, x ; v_mul_f32 x, o, o", w);
    run<0, 8>("v_rsq_f32 y, x ; v_mov_b32 x, other", w);
    run<1, 8>("v_rsq_f32 y, x ; v_pk_mul_f32 {x, x'}, o, o", w);
    run<2, 8>("v_rsq_f32 y, x ; v_mul_f32 x, o, o", w);
    run<3, 8>("v_rsq_f32 y, x ; s_nop 0 ; v_pk_mul_f32 {x, x'}, o, o", w);
    run<4, 8>("v_rsq_f32 y, x ; v_add_f32 (independent) ; v_mov_b32 x, other", w);
  }
  for (int w = 1; w <= 4; w *= 2) {
    run2<0, true>(w); run2<1, true>(w); run2<2, true>(w); run2<3, true>(w);
    run2<0, false>(w); run2<1, false>(w); run2<2, false>(w); run2<3, false>(w);
  }
  return 0;
}
