// Decision microbenchmark: one target per lane (the small-problem kernels) is co-limited by LDS bandwidth — every lane receives the
// 32-byte source record per pair (2 x ds_read_b128 = 16 LDS cycles per wave-pair against 64 VALU cycles per SIMD, 4 SIMDs per LDS).
// A source is wave-uniform data: read through the SCALAR cache (s_load into SGPRs, used as the one scalar operand a VALU
// instruction may have) it costs neither LDS bandwidth nor vector registers nor a staging pass.  Compares, on 2^14 x 2^14 Laplace SL fp64
// split 16 ways (the shipped plan): the library's LDS-tiled kernel against a kernel with scalar-fed sources; checks that the partial
// sums agree.   Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -o smem_sources smem_sources.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "sctl_amd/device/eval_kernel.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace sctl_amd;

template <int U>
__global__ void __launch_bounds__(kBlock) smem_kernel(const EvalArgs<double> a) {
  using Ker = Laplace3D_FxU;
  const typename Ker::Consts<double> K(nullptr);
  const int tid = threadIdx.x;
  int64_t t = (int64_t)blockIdx.x * kBlock + tid;
  if (t >= a.Nt) t = a.Nt - 1;
  const double xt[3] = {a.xt[t * 3], a.xt[t * 3 + 1], a.xt[t * 3 + 2]};
  const int64_t s_begin = (int64_t)blockIdx.y * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  // constant address space: the data are not written while the kernel runs, and a wave-uniform address then becomes a scalar load
  typedef const double __attribute__((address_space(4))) * cptr;
  const cptr xs = (cptr)(uintptr_t)a.xs, f = (cptr)(uintptr_t)a.f;
  double acc[1] = {0};
  int64_t s = s_begin;
  for (; s + U <= s_end; s += U) {
    double tacc[1] = {0};
#pragma unroll
    for (int u = 0; u < U; u++) {
      const double rec[4] = {xs[(s + u) * 3], xs[(s + u) * 3 + 1], xs[(s + u) * 3 + 2], f[s + u]};   // wave-uniform addresses: scalar loads
      const double d[3] = {xt[0] - rec[0], xt[1] - rec[1], xt[2] - rec[2]};
      Ker::pair<double, 2, false>(tacc, d, rec, a.ctx, K);
    }
    if (!(fabs(tacc[0]) <= 1.7976931348623157e308)) {   // a coincident pair in this group: redo it masked
      tacc[0] = 0;
      for (int u = 0; u < U; u++) {
        const double rec[4] = {xs[(s + u) * 3], xs[(s + u) * 3 + 1], xs[(s + u) * 3 + 2], f[s + u]};
        const double d[3] = {xt[0] - rec[0], xt[1] - rec[1], xt[2] - rec[2]};
        Ker::pair<double, 2, true>(tacc, d, rec, a.ctx, K);
      }
    }
    acc[0] += tacc[0];
  }
  for (; s < s_end; s++) {
    const double rec[4] = {xs[s * 3], xs[s * 3 + 1], xs[s * 3 + 2], f[s]};
    const double d[3] = {xt[0] - rec[0], xt[1] - rec[1], xt[2] - rec[2]};
    Ker::pair<double, 2, true>(acc, d, rec, a.ctx, K);
  }
  const int64_t tt = (int64_t)blockIdx.x * kBlock + tid;
  if (tt < a.Nt) a.partial[(int64_t)blockIdx.y * a.Nt + tt] = acc[0];
}

int main() {
  CHECK(hipSetDevice(0));
  for (int logn : {13, 14, 15}) {
    const int64_t N = 1ll << logn;
    std::vector<double> hx(N * 3), hs(N * 3), hf(N);
    srand48(1);
    for (auto& v : hx) v = drand48();
    for (auto& v : hs) v = drand48();
    for (auto& v : hf) v = drand48() - 0.5;
    const int splits = (logn == 13) ? 32 : 16;
    double *xt, *xs, *f, *pa, *pb, *v;
    CHECK(hipMalloc(&xt, N * 24)); CHECK(hipMalloc(&xs, N * 24)); CHECK(hipMalloc(&f, N * 8)); CHECK(hipMalloc(&v, N * 8));
    CHECK(hipMalloc(&pa, N * 8 * splits)); CHECK(hipMalloc(&pb, N * 8 * splits));
    CHECK(hipMemcpy(xt, hx.data(), N * 24, hipMemcpyHostToDevice)); CHECK(hipMemcpy(xs, hs.data(), N * 24, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(f, hf.data(), N * 8, hipMemcpyHostToDevice));
    EvalArgs<double> a{};
    a.Nt = N; a.Ns = N; a.xt = xt; a.xs = xs; a.xn = nullptr; a.f = f; a.v_trg = v; a.chunk = N / splits; a.scale = 1;
    const dim3 grid((unsigned)(N / kBlock), (unsigned)splits);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
      for (int i = 0; i < 20; i++) launch();
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 200; i++) launch();
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 200 * 1e3;
    };
    a.partial = pa;
    const double t_lds = time([&] { hipLaunchKernelGGL((eval_kernel<Laplace3D_FxU, double, 2, 1>), grid, dim3(kBlock), 0, 0, a); });
    a.partial = pb;
    const EvalArgs<double> b = a;
    const double t_s4 = time([&] { hipLaunchKernelGGL((smem_kernel<4>), grid, dim3(kBlock), 0, 0, b); });
    std::vector<double> ra(N * splits), rb(N * splits);
    CHECK(hipMemcpy(rb.data(), pb, N * 8 * splits, hipMemcpyDeviceToHost));
    const double t_s8 = time([&] { hipLaunchKernelGGL((smem_kernel<8>), grid, dim3(kBlock), 0, 0, b); });
    const double t_s16 = time([&] { hipLaunchKernelGGL((smem_kernel<16>), grid, dim3(kBlock), 0, 0, b); });
    CHECK(hipMemcpy(ra.data(), pa, N * 8 * splits, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < ra.size(); i++) { num += (ra[i] - rb[i]) * (ra[i] - rb[i]); den += ra[i] * ra[i]; }
    printf("2^%d x 2^%d, %d splits: LDS-tiled %7.1f us | scalar-fed sources, groups of 4: %7.1f us, of 8: %7.1f us, of 16: %7.1f us | rel-L2 of the partial sums %.2e\n", logn, logn,
           splits, t_lds, t_s4, t_s8, t_s16, sqrt(num / den));
    CHECK(hipFree(xt)); CHECK(hipFree(xs)); CHECK(hipFree(f)); CHECK(hipFree(v)); CHECK(hipFree(pa)); CHECK(hipFree(pb));
  }
  return 0;
}
