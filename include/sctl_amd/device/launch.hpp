// Launch tables: one KernelEntry per micro-kernel, filled in its own translation unit (inst_*.hip) so the
// ~200 kernel instantiations compile in parallel.  capi.hip only sees function pointers.
#pragma once
#include "eval_kernel.hpp"
#include "lists_kernel.hpp"

namespace sctl_amd {

constexpr int kNumT = 3;                       // targets per lane: 1, 2, 4
constexpr int kTvalues[kNumT] = {1, 2, 4};
constexpr int kNumMode = 3;                    // rsqrt refinement: seed, Newton, Halley (ukernels.hpp)

template <class R> using EvalLaunch = void (*)(const EvalArgs<R>&, dim3 grid, hipStream_t);
template <class R> using MatrixBatchLaunch = void (*)(const MatTile* tiles, int64_t ntiles, const R* xt, const R* xs, const R* xn, R* M, R scale, const KerCtx&, hipStream_t);
template <class R> using ListsLaunch = void (*)(const ListArgs<R>&, int64_t nblocks, hipStream_t);
template <class R> using MatrixLaunch = void (*)(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, R* M, R scale, const KerCtx&, dim3 grid, hipStream_t);

struct KernelEntry {
  const char* name;
  int id, k0, k1, nd, flops, nrec, ctx_bytes;
  double scale;
  double acc_factor[kNumMode];   // pair() of mode m accumulates acc_factor[m] x the kernel value: the launch scale is scale / acc_factor[m]
  EvalLaunch<double> eval_f64[kNumMode][kNumT];
  EvalLaunch<float> eval_f32[kNumMode][kNumT];     // modes 0 and 1 only (mode 2 aliases mode 1)
  MatrixLaunch<double> matrix_f64[kNumMode];
  MatrixLaunch<float> matrix_f32[kNumMode];
  MatrixBatchLaunch<double> matrix_batch_f64[kNumMode];
  MatrixBatchLaunch<float> matrix_batch_f32[kNumMode];
  ListsLaunch<double> lists_f64[kNumMode];          // lists_kernel.hpp
  ListsLaunch<float> lists_f32[kNumMode];
};

template <class Ker, class R, int MODE, int T> void launch_eval(const EvalArgs<R>& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((eval_kernel<Ker, R, MODE, T>), grid, dim3(kBlock), 0, st, a);
}
template <class Ker, class R, int MODE> void launch_matrix(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, R* M, R scale,
                                                           const KerCtx& ctx, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((matrix_kernel<Ker, R, MODE>), grid, dim3(kBlock), 0, st, Nt, Ns, xt, xs, xn, M, scale, ctx);
}

template <class Ker, class R, int MODE> void launch_matrix_batch(const MatTile* tiles, int64_t ntiles, const R* xt, const R* xs, const R* xn, R* M, R scale,
                                                                 const KerCtx& ctx, hipStream_t st) {
  hipLaunchKernelGGL((matrix_batch_kernel<Ker, R, MODE>), dim3((unsigned)ntiles), dim3(kBlock), 0, st, tiles, xt, xs, xn, M, scale, ctx);
}

template <class Ker, class R, int MODE> void launch_lists(const ListArgs<R>& a, int64_t nblocks, hipStream_t st) {
  hipLaunchKernelGGL((lists_kernel<Ker, R, MODE>), dim3((unsigned)nblocks), dim3(kListWave), 0, st, a);
}

template <class Ker> KernelEntry make_entry(int ctx_bytes) {
  KernelEntry e{};
  e.name = Ker::NAME; e.id = Ker::ID; e.k0 = Ker::K0; e.k1 = Ker::K1; e.nd = Ker::ND; e.flops = Ker::FLOPS; e.nrec = Ker::NREC;
  e.ctx_bytes = ctx_bytes; e.scale = Ker::scale();
  for (int m = 0; m < kNumMode; m++) e.acc_factor[m] = Ker::acc_factor(m);
#define SCTL_AMD_ROW(R, arr, M, MM) \
  arr[M][0] = launch_eval<Ker, R, MM, 1>; arr[M][1] = launch_eval<Ker, R, MM, 2>; arr[M][2] = launch_eval<Ker, R, MM, 4>;
  SCTL_AMD_ROW(double, e.eval_f64, 0, 0) SCTL_AMD_ROW(double, e.eval_f64, 1, 1) SCTL_AMD_ROW(double, e.eval_f64, 2, 2)
  SCTL_AMD_ROW(float, e.eval_f32, 0, 0) SCTL_AMD_ROW(float, e.eval_f32, 1, 1) SCTL_AMD_ROW(float, e.eval_f32, 2, 1)
#undef SCTL_AMD_ROW
  e.matrix_f64[0] = launch_matrix<Ker, double, 0>; e.matrix_f64[1] = launch_matrix<Ker, double, 1>; e.matrix_f64[2] = launch_matrix<Ker, double, 2>;
  e.matrix_f32[0] = launch_matrix<Ker, float, 0>; e.matrix_f32[1] = launch_matrix<Ker, float, 1>; e.matrix_f32[2] = launch_matrix<Ker, float, 1>;
  e.matrix_batch_f64[0] = launch_matrix_batch<Ker, double, 0>; e.matrix_batch_f64[1] = launch_matrix_batch<Ker, double, 1>;
  e.matrix_batch_f64[2] = launch_matrix_batch<Ker, double, 2>;
  e.matrix_batch_f32[0] = launch_matrix_batch<Ker, float, 0>; e.matrix_batch_f32[1] = launch_matrix_batch<Ker, float, 1>;
  e.matrix_batch_f32[2] = launch_matrix_batch<Ker, float, 1>;
  e.lists_f64[0] = launch_lists<Ker, double, 0>; e.lists_f64[1] = launch_lists<Ker, double, 1>; e.lists_f64[2] = launch_lists<Ker, double, 2>;
  e.lists_f32[0] = launch_lists<Ker, float, 0>; e.lists_f32[1] = launch_lists<Ker, float, 1>; e.lists_f32[2] = launch_lists<Ker, float, 1>;
  return e;
}

// defined in inst_*.hip
const KernelEntry& entry_Laplace3D_FxU();
const KernelEntry& entry_Laplace3D_DxU();
const KernelEntry& entry_Laplace3D_FxdU();
const KernelEntry& entry_Stokes3D_FxU();
const KernelEntry& entry_Stokes3D_DxU();
const KernelEntry& entry_Stokes3D_FxT();
const KernelEntry& entry_Stokes3D_FSxU();
const KernelEntry& entry_Stokes3D_FxUP();
const KernelEntry& entry_Laplace3D_FDxUdU();
const KernelEntry& entry_Helmholtz3D_FxU();

}  // namespace sctl_amd
