// Exploration tool: where the 24-bit seed of 1/sqrt(r2) comes from in the tile-centred Laplace far loop (fp64, full precision).
// The shipped loop is 4 (distance) + v_rsq_f64 (16 cycles = 4 issue slots) + 5 (Halley) + 1 (accumulate) = 14 slots per pair.
// v_rsq_f32 costs half of v_rsq_f64 (tools/ubench/valu_rates.hip) and 32-bit integer instructions a fraction of an fp64 one, so a seed
// taken through fp32 BITS — no v_cvt in either direction — could cost ~3 slots instead of 4:
//   SEED 0  v_rsq_f64                                                                     (shipped)
//   SEED 1  v_cvt_f32_f64, v_rsq_f32, v_cvt_f64_f32
//   SEED 2  high word of r2 -> fp32 bits by ONE v_lshl_add_u32, v_rsq_f32, v_cvt_f64_f32
//   SEED 3  as 2 on the way in; on the way out the fp32 bits become the HIGH word of a double by ONE v_alignbit_b32 (shift by 3, the top
//           three exponent bits OR-ed in); the low word is left undefined (it is below the seed's accuracy).  An OR cannot add the
//           exponent re-bias 896 = 0x380 (bit 7 of it collides with the fp32 exponent), only 0x200, 0x400 or 0x600: with 0x400 the seed is
//           2^128 / r, which the loop absorbs by working on coordinates scaled by 2^-128 (r2 scaled by 2^-256, so that e = 1 - r2 y^2 is
//           unscaled) and a result scaled by 2^128.
// The fp32-bit seeds are good to ~2^-19.5 (three mantissa bits dropped each way); the Halley step's residual 5/16 e^3, e = 2 d, stays below 2^-56.
// Not part of the product; results recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int SEED> __device__ __forceinline__ double seed(double r2, unsigned cin, unsigned cout) {
  if (SEED == 0) return __builtin_amdgcn_rsq(r2);
  if (SEED == 1) return (double)__builtin_amdgcn_rsqf((float)r2);
  unsigned fb;                                                               // fp32 bits of r2 (scaled), mantissa truncated to 20 bits
  asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(fb) : "v"(__double2hiint(r2)), "s"(cin));   // (the compiler widens (hi << 3) + c to the 64-bit value: three instructions)
  const float yf = __builtin_amdgcn_rsqf(__uint_as_float(fb));
  if (SEED == 2) return (double)yf;
  unsigned lo;
  asm("" : "=v"(lo) : "v"(yf));                                                      // any bits will do: they sit below 2^-20 of the seed (a frozen poison value becomes 0 and costs a move)
  const unsigned hi = __builtin_amdgcn_alignbit(cout, __float_as_uint(yf), 3);   // (bits >> 3) | (cout << 29)
  return __hiloint2double((int)hi, (int)lo);
}

// MODE 2 Halley, MODE 1 Newton (normalised)
template <int SEED, int MODE> __device__ __forceinline__ void far_pair(double& acc, const double (&m2x)[3], double tt, d2 a, d2 b, double f, double c38, unsigned cin,
                                                                       unsigned cout) {
  const double r2 = __builtin_fma(m2x[0], a[0], __builtin_fma(m2x[1], a[1], __builtin_fma(m2x[2], b[0], tt + b[1])));
  double y = seed<SEED>(r2, cin, cout);
  const double ay = r2 * y;
  const double e = __builtin_fma(-ay, y, 1.0);
  const double ye = y * e;
  if (MODE == 1) y = __builtin_fma(ye, 0.5, y);
  else y = __builtin_fma(ye, __builtin_fma(e, c38, 0.5), y);
  acc = __builtin_fma(f, y, acc);
}

template <int SEED, int MODE, int T, int U>
__global__ void __launch_bounds__(64) lap(const double* __restrict__ xt, const double* __restrict__ xs, const double* __restrict__ fs, double* __restrict__ out, int Ns,
                                          double cs, double outscale) {
  constexpr int TILE = 64;
  __shared__ d2 tile[TILE * 2];
  __shared__ double dens[TILE];
  const int tid = threadIdx.x;
  double c38 = 0.375; asm volatile("" : "+v"(c38));
  // 2: plain fp32 bits of r2;  3: r2 carries 2^-256, the fp32 value is the unscaled one (both constants are mod 2^32)
  unsigned cin = SEED == 3 ? 0xC0000000u : 0x40000000u, cout = 2u;
  asm volatile("" : "+s"(cin), "+s"(cout));
  double m2x[T][3], tt[T], acc[T];
#pragma unroll
  for (int j = 0; j < T; j++) {
    const long t = (long)blockIdx.x * (64 * T) + j * 64 + tid;
    double x[3];
    for (int k = 0; k < 3; k++) x[k] = (xt[t * 3 + k] - 0.5) * cs;
    tt[j] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    for (int k = 0; k < 3; k++) m2x[j][k] = -2 * x[k];
    acc[j] = 0;
  }
  for (int s0 = 0; s0 < Ns; s0 += TILE) {
    __syncthreads();
    {
      const long s = s0 + tid;
      const double x = (xs[s * 3] + 1.0) * cs, y = (xs[s * 3 + 1] + 1.0) * cs, z = (xs[s * 3 + 2] + 1.0) * cs;   // sources in [1,2)^3: far from the targets' cube about 0
      tile[tid * 2] = d2{x, y};
      tile[tid * 2 + 1] = d2{z, x * x + y * y + z * z};
      dens[tid] = fs[s];
    }
    __syncthreads();
#pragma unroll U
    for (int s = 0; s < TILE; s++) {
      const d2 a = tile[s * 2], b = tile[s * 2 + 1];
      const double f = dens[s];
#pragma unroll
      for (int j = 0; j < T; j++) far_pair<SEED, MODE>(acc[j], m2x[j], tt[j], a, b, f, c38, cin, cout);
    }
  }
#pragma unroll
  for (int j = 0; j < T; j++) out[(long)blockIdx.x * (64 * T) + j * 64 + tid] = acc[j] * outscale;
}

struct Var { const char* name; void (*fn)(const double*, const double*, const double*, double*, int, double, double); int T; double cs, outscale; };
#define V(SEED, MODE, T, U) Var{"SEED=" #SEED " MODE=" #MODE " T=" #T " U=" #U, lap<SEED, MODE, T, U>, T, SEED == 3 ? 0x1p-128 : 1.0, SEED == 3 ? 0x1p-128 : 1.0}

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
  std::vector<Var> vars = {V(0, 2, 2, 4), V(1, 2, 2, 4), V(2, 2, 2, 4), V(3, 2, 2, 4), V(0, 1, 2, 4), V(3, 1, 2, 4), V(0, 2, 2, 2), V(3, 2, 2, 2), V(3, 2, 2, 8)};
  std::vector<double> ref(N), got(N);
  for (size_t i = 0; i < vars.size(); i++) {
    auto& v = vars[i];
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(N / (64 * v.T)), dim3(64), 0, 0, xt, xs, f, out, N, v.cs, v.outscale);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(got.data(), out, N * 8, hipMemcpyDeviceToHost));
    if (i == 0) ref = got;
    double d = 0, n = 0, mx = 0; for (int k = 0; k < N; k++) { d += (got[k] - ref[k]) * (got[k] - ref[k]); n += ref[k] * ref[k]; mx = fmax(mx, fabs(got[k] - ref[k]) / fabs(ref[k])); }
    const double pps = (double)N * N / (best * 1e-3);
    printf("%-28s %8.2f ms  %.3e pairs/s  %5.1f%% of 78.6 TF   rel-L2 vs first %.2e  max rel %.2e\n", v.name, best, pps, pps * 11 / 78.6e12 * 100, sqrt(d / n), mx);
    fflush(stdout);
  }
  return 0;
}
