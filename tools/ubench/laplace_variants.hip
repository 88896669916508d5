// Exploration tool: variants of the Laplace single-layer fp64 pair loop (targets per lane, unroll, LDS tile size,
// mask policy, refinement) timed on a 2^20 x 2^20 problem.  Not part of the product; results recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// MASK: 0 = hi-word compare/select on the seed, 1 = none (WRONG for coincident points; upper bound only),
//       2 = fp64 compare of r2 + select on the seed's high word, 3 = branch (exec mask) around the pair
// MODE: 0 seed, 1 Newton, 2 Halley
template <int MASK, int MODE> __device__ __forceinline__ void pair(double& acc, double dx, double dy, double dz, double f, double c38) {
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  if (MASK == 3) { if (!(r2 > 0)) return; }
  double y = __builtin_amdgcn_rsq(r2);
  if (MASK == 0) {
    int hi = __double2hiint(y);
    asm("" : "+v"(hi));
    hi = (hi == 0x7ff00000) ? 0 : hi;
    y = __hiloint2double(hi, __double2loint(y));
  } else if (MASK == 2) {
    int hi = __double2hiint(y);
    hi = (r2 > 0) ? hi : 0;
    y = __hiloint2double(hi, __double2loint(y));
  }
  if (MODE >= 1) {
    const double a = r2 * y;
    const double e = __builtin_fma(-a, y, 1.0);
    const double ye = y * e;
    if (MODE == 1) y = __builtin_fma(ye, 0.5, y);
    else y = __builtin_fma(ye, __builtin_fma(e, c38, 0.5), y);
  }
  acc = __builtin_fma(f, y, acc);
}

template <int T, int U, int TILE, int MASK, int MODE, int MINW>
__global__ void __launch_bounds__(256, MINW) lap(const double* __restrict__ xt, const double* __restrict__ xs, const double* __restrict__ fs,
                                                double* __restrict__ out, int Nt, int Ns) {
  __shared__ d2 tile[TILE * 2];
  const int tid = threadIdx.x;
  double c38 = 0.375; asm volatile("" : "+v"(c38));
  double x[T][3], acc[T];
#pragma unroll
  for (int j = 0; j < T; j++) {
    const long t = (long)blockIdx.x * (256 * T) + j * 256 + tid;
    for (int k = 0; k < 3; k++) x[j][k] = xt[t * 3 + k];
    acc[j] = 0;
  }
  for (int s0 = 0; s0 < Ns; s0 += TILE) {
    __syncthreads();
    for (int i = tid; i < TILE; i += 256) {
      const long s = s0 + i;
      tile[i * 2] = d2{xs[s * 3], xs[s * 3 + 1]};
      tile[i * 2 + 1] = d2{xs[s * 3 + 2], fs[s]};
    }
    __syncthreads();
#pragma unroll U
    for (int s = 0; s < TILE; s++) {
      const d2 a = tile[s * 2], b = tile[s * 2 + 1];
#pragma unroll
      for (int j = 0; j < T; j++) pair<MASK, MODE>(acc[j], x[j][0] - a[0], x[j][1] - a[1], x[j][2] - b[0], b[1], c38);
    }
  }
#pragma unroll
  for (int j = 0; j < T; j++) out[(long)blockIdx.x * (256 * T) + j * 256 + tid] = acc[j] * 0.07957747154594767;
}

struct Var { const char* name; void (*fn)(const double*, const double*, const double*, double*, int, int); int T; };
#define V(T, U, TILE, MASK, MODE, MINW) Var{"T=" #T " U=" #U " TILE=" #TILE " MASK=" #MASK " MODE=" #MODE " MINW=" #MINW, lap<T, U, TILE, MASK, MODE, MINW>, T}

int main() {
  const int N = 1 << 20;
  std::vector<double> h(N * 3); for (auto& v : h) v = drand48();
  double *xt, *xs, *f, *out;
  CHECK(hipMalloc(&xt, N * 24)); CHECK(hipMalloc(&xs, N * 24)); CHECK(hipMalloc(&f, N * 8)); CHECK(hipMalloc(&out, N * 8));
  CHECK(hipMemcpy(xt, h.data(), N * 24, hipMemcpyHostToDevice));
  for (auto& v : h) v = drand48();
  CHECK(hipMemcpy(xs, h.data(), N * 24, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(f, h.data(), N * 8, hipMemcpyHostToDevice));
  std::vector<Var> vars = {
    V(2, 2, 256, 0, 2, 1), V(2, 2, 256, 1, 2, 1), V(2, 2, 256, 2, 2, 1), V(2, 2, 256, 3, 2, 1),
    V(1, 4, 256, 0, 2, 1), V(4, 1, 256, 0, 2, 1), V(4, 2, 256, 0, 2, 1), V(2, 4, 256, 0, 2, 1), V(2, 1, 256, 0, 2, 1),
    V(2, 2, 1024, 0, 2, 1), V(4, 1, 1024, 0, 2, 1), V(2, 2, 256, 0, 2, 2), V(4, 1, 256, 0, 2, 2), V(4, 1, 256, 1, 2, 1),
    V(2, 2, 256, 0, 1, 1), V(2, 2, 256, 0, 0, 1), V(2, 2, 256, 1, 0, 1), V(8, 1, 256, 0, 2, 1),
  };
  std::vector<double> ref(N), got(N);
  for (size_t i = 0; i < vars.size(); i++) {
    auto& v = vars[i];
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(N / (256 * v.T)), dim3(256), 0, 0, xt, xs, f, out, N, N);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(got.data(), out, N * 8, hipMemcpyDeviceToHost));
    if (i == 0) ref = got;
    double d = 0, n = 0; for (int k = 0; k < N; k++) { d += (got[k] - ref[k]) * (got[k] - ref[k]); n += ref[k] * ref[k]; }
    const double pps = (double)N * N / (best * 1e-3);
    printf("%-44s %8.2f ms  %.3e pairs/s  %5.1f%% of 78.6 TF   rel-L2 vs first %.2e\n", v.name, best, pps, pps * 11 / 78.6e12 * 100, sqrt(d / n));
    fflush(stdout);
  }
  return 0;
}
