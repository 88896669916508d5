// Host driver of the tile-centred Laplace single-layer path (centered_kernel.hpp): Morton-sort the targets on the device
// (rocPRIM radix sort of 63-bit keys), evaluate on the sorted order, scatter-add the result back.  Everything is enqueued
// on the caller's stream; temporaries are carved out of the stream's scratch block (workspace.hpp).
#include "centered_kernel.hpp"
#include "centered_mfma_kernel.hpp"
#include <sctl_amd/device/launch.hpp>
#include "workspace.hpp"

#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>

namespace sctl_amd {

#define CENTERED_TRY(expr)            \
  do {                                \
    hipError_t e_ = (expr);           \
    if (e_ != hipSuccess) return e_;  \
  } while (0)

// Targets per lane of the vector-pipe kernel (centered_kernel.hpp): 64 T Morton-consecutive targets share a centre and every staged tile.  fp64 single and double
// layer: FOUR — half the per-tile staging and LDS reads per pair for a somewhat larger cluster; the full-precision kernel still fits four waves per SIMD (125
// registers), the others three.  A/B on one box (profiles/r03_ab_centered_T.txt; T = 2 -> 4 -> 8): Laplace SL 2^20 x 2^20 405.4 -> 397.7 -> 405.9 ms at full
// precision, 383.6 -> 369.5 -> 379.4 ms at 10 digits; double layer 563.9 -> 542.5 -> 534.1 ms.  fp32 keeps two: its far pairs are written as ONE packed stream over
// exactly two targets, and its default accuracy runs on the matrix cores anyway.  The gradient kernel (round 4) takes three.  Each policy says so itself (targets_per_lane).

namespace {
// fp32 Laplace single and double layer at the seed's accuracy take the kernel whose contractions run on the bf16 matrix cores
// (centered_mfma_kernel.hpp); SCTL_AMD_MFMA_F32=0 keeps the packed-VALU kernel (A/B runs and tests that compare the two)
bool use_mfma_f32() {
  const char* e = std::getenv("SCTL_AMD_MFMA_F32");
  return !(e && e[0] == '0');
}
}  // namespace
// 2 when the tile-centred path of (kernel, real, mode) takes its far distances from the bf16 matrix cores, 1 when it runs on the vector pipe alone
// (what sctl_amd_eval_pipe reports; must say what launch_centered does)
int centered_pipe(int kernel_id, int real, int mode) {
  const bool has = kernel_id == Laplace3D_FxU::ID || kernel_id == Laplace3D_DxU::ID || kernel_id == Stokes3D_FxU::ID || kernel_id == Stokes3D_FSxU::ID ||
                   kernel_id == Stokes3D_FxUP::ID || kernel_id == Stokes3D_DxU::ID || kernel_id == Stokes3D_FxT::ID || kernel_id == Laplace3D_FxdU::ID || kernel_id == Laplace3D_FDxUdU::ID;   // (Stokeslet family and stresslet, round 4: r.f, r.n are further contractions, as the double layer's numerator)
  return (has && real == 1 /* SCTL_AMD_F32 */ && mode == 0 && use_mfma_f32()) ? 2 : 1;
}
// Targets per wave (= per workgroup) of that path: 64 per target of a lane on the vector pipe (the policy's targets_per_lane); for the matrix-core kernels 256 = eight
// 32-column blocks — the per-tile staging (one or two contraction rows per source) is shared by twice the pairs of the 128-target form — except for the single layer on a
// SPARSE target set: below 2^20 targets in the domain (`density`: the size of the set the targets were cut from) the larger cluster of 256 targets makes too many sources
// "near", and the 128-target form (120 registers, four waves per SIMD) wins.  A/B on one box, fp32 (tools/ab_mfma_cb_sizes.py, profiles/r04_ab_mfma_cb_sizes.txt), 256 vs 128
// targets per wave: single layer 2^18 8.27 vs 7.75 ms, 2^19 29.2 vs 28.8, 2^20 111.4 vs 112.0, 2^21 427.4 vs 438.0; double layer 12.6 vs 13.6, 46.3 vs 50.8, 175.6 vs
// 198.1, 679 vs 773 (256 everywhere).  SCTL_AMD_MFMA_CB=4 / 8 overrides (A/B runs, tests).
int centered_targets_per_wave(int kernel_id, int real, int mode, int64_t density) {
  if (centered_pipe(kernel_id, real, mode) != 2) {
    if (kernel_id == Laplace3D_FxdU::ID) return 64 * CenteredFxdU<double>::targets_per_lane<double>();
    if (kernel_id == Stokes3D_FxUP::ID) return 64 * CenteredStokeslet<double, Stokes3D_FxUP>::targets_per_lane<double>();
    return 64 * (real == 0 /* SCTL_AMD_F64 */ ? CenteredFxU<double>::targets_per_lane<double>() : CenteredFxU<float>::targets_per_lane<float>());
  }
  if (const char* e = (kernel_id == Laplace3D_FxU::ID || kernel_id == Laplace3D_DxU::ID) ? std::getenv("SCTL_AMD_MFMA_CB") : nullptr) {
    if (e[0] == '8') return 256;
    if (e[0] == '4') return 128;
  }
  if (kernel_id != Laplace3D_FxU::ID && kernel_id != Laplace3D_DxU::ID) return 128;   // the Stokeslet family, the stresslet: four column blocks
  return (kernel_id == Laplace3D_FxU::ID && density < ((int64_t)1 << 20)) ? 128 : 256;
}
namespace {
// fp32 kernels with several outputs per target: the policy of centered_mfma_moments_f32_kernel that serves a tile-centred policy (void: none)
template <class R> struct CenteredStresslet { using Ker = Stokes3D_DxU; };   // (fp32 only: a tag for eval_centered_t; the stresslet has no vector-pipe centred form)
template <class CP> struct mfma_moments_policy { using type = void; };
template <class KER> struct mfma_moments_policy<CenteredStokeslet<float, KER>> { using type = MfmaStokeslet<KER>; };
template <> struct mfma_moments_policy<CenteredStresslet<float>> { using type = MfmaStresslet; };
template <class R> struct CenteredTraction { using Ker = Stokes3D_FxT; };
template <class R> struct CenteredFusedLaplace { using Ker = Laplace3D_FDxUdU; };
template <> struct mfma_moments_policy<CenteredFusedLaplace<float>> { using type = MfmaFusedLaplace; };
template <> struct mfma_moments_policy<CenteredFxdU<float>> { using type = MfmaGradient; };
template <> struct mfma_moments_policy<CenteredTraction<float>> { using type = MfmaTraction; };
template <class CP, class R, int MODE> void launch_centered(const EvalArgs<R>& a, dim3 grid, int per_wave, hipStream_t st) {
  using MP = typename mfma_moments_policy<CP>::type;
  if constexpr (std::is_same<R, float>::value && !std::is_void<MP>::value) {   // the matrix-core moments kernel at the seed's accuracy, nothing else
    if constexpr (MODE == 0) hipLaunchKernelGGL((centered_mfma_moments_f32_kernel<MP>), grid, dim3(kWaveBlock), 0, st, a);   // (capi.hip asks for this path at MODE 0 only)
    return;
  } else {
  if constexpr (std::is_same<R, float>::value && MODE == 0) {
    if (use_mfma_f32()) {   // (the caller sized grid.x with per_wave = centered_targets_per_wave)
      constexpr bool DL = std::is_same<CP, CenteredDxU<float>>::value;
      if (per_wave == 256) {
        if constexpr (DL) hipLaunchKernelGGL((centered_mfma_f32_kernel<true, 8>), grid, dim3(kWaveBlock), 0, st, a);
        else hipLaunchKernelGGL(centered_mfma_fxu256_f32_kernel, grid, dim3(kWaveBlock), 0, st, a);
      } else {
        hipLaunchKernelGGL((centered_mfma_f32_kernel<DL, 4>), grid, dim3(kWaveBlock), 0, st, a);
      }
      return;
    }
  }
  hipLaunchKernelGGL((centered_kernel<CP, R, MODE, CP::template targets_per_lane<R>()>), grid, dim3(kWaveBlock), 0, st, a);
  }
}
}  // namespace

// v_trg[Nt] += scale * sum_s (kernel of the policy CP), mode = rsqrt refinement (ukernels.hpp)
// Launch geometry of the centred kernel: one wave per workgroup, 64*T targets each.  The kernel holds 103 VGPRs, so 16
// waves are resident per CU; the work per wave varies with its share of near sources (0.5 % .. 34 % at 2^20 uniform
// points), so the source range is split until there are >= 64 "rounds" of workgroups (counted at 128 targets per wave) — measured on 2^20 x 2^20:
// 1 split 517 ms, 4 splits 480 ms, 16 splits 469 ms (exact kernel on the same GPU: 520 ms).
// Second rule (round 2): a split's source data (src_bytes per source: coordinates, density, normals as the kernel reads them) should fit the
// 4 MB L2 of the XCD that owns the split (centered_kernel.hpp), i.e. <= 2 MB: at 2^23 fp32 sources in 2 splits every wave streamed 64 MB per
// split through a 4 MB cache and the launch pulled 3.66 TB through the fabric (13 600 x the algorithmic bytes; PMC, profiles/r02_laplace_sl_f32_*).
void centered_plan(int64_t Nt, int64_t Ns, int cus, int src_bytes, int out_bytes, int* T, int* splits, int64_t* chunk) {
  *T = 2;   // (the split rule was sized for 128 targets per wave; one target per lane measured 498 ms against 465 ms at 2^20, round 1)
  const int64_t wg_x = (Nt + kWaveBlock * 2 - 1) / (kWaveBlock * 2);
  const int64_t want = (int64_t)cus * 16 * 64;   // (x 32 until round 4: 2^20 x 2^20 in 32 instead of 16 splits — workgroups half as long, a shorter last round — 402.8 -> 398.5 ms,
                                                 //  at 10 digits 373.4 -> 369.7; 64 splits the same again: profiles/r04_ab_rank_splits.txt)
  const int64_t ntile = (Ns + kWaveTile - 1) / kWaveTile;
  int64_t s = (want + wg_x - 1) / wg_x;
  const int64_t s_l2 = (Ns * src_bytes + (2 << 20) - 1) / (2 << 20);
  if (s < s_l2) s = s_l2;
  if (s > ntile / 64) s = ntile / 64;      // at least 64 tiles (4096 sources) per split
  if (s > 8) s = (s + 7) & ~(int64_t)7;    // the XCD-aware mapping needs a multiple of 8
  if (s > 256) s = 256;                    // (64 until round 4: a rank's slab of 2^17 targets then ran 8 rounds of 6 ms workgroups and lost 4 % to the last one — 51.7 -> 50.1 ms with 128 splits,
                                           //  95.5 -> 98.5 % of an eighth of the whole problem: profiles/r04_ab_rank_splits.txt)
  {   // ... and at most 2 GB of partial sums (out_bytes per target and split), in eights while that leaves any
    const int64_t fit = ((int64_t)2 << 30) / (Nt * out_bytes > 0 ? Nt * out_bytes : 1);
    if (s > fit) s = (fit >= 8) ? (fit & ~(int64_t)7) : fit;
  }
  if (s < 1) s = 1;
  const int64_t tiles_per = (ntile + s - 1) / s;
  *chunk = tiles_per * kWaveTile;
  *splits = (int)((Ns + *chunk - 1) / *chunk);
}

template <class CP, class R>
hipError_t eval_centered_t(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, const R* f, R* v_trg, double scale_d, int mode, int cus,
                           hipStream_t st, bool presorted, int64_t density) {
  const R scale = (R)scale_d;
  constexpr int K1 = CP::Ker::K1;
  int T, splits;
  int64_t chunk;
  centered_plan(Nt, Ns, cus, (int)sizeof(R) * (3 + CP::Ker::ND + CP::Ker::K0), (int)sizeof(R) * K1, &T, &splits, &chunk);
  if (presorted) {   // the caller keeps the targets in Morton order (sctl_amd_op_*): no sort, no gather, results in place
    R* partial = nullptr;
    if (splits > 1) {
      void* base = nullptr;
      CENTERED_TRY(workspace_acquire(st, sizeof(R) * (size_t)splits * (size_t)Nt * K1, &base));
      partial = (R*)base;
    }
    EvalArgs<R> a{};
    a.Nt = Nt; a.Ns = Ns; a.xt = xt; a.xs = xs; a.xn = xn; a.f = f; a.v_trg = v_trg; a.partial = partial;
    a.chunk = chunk; a.scale = scale;
    a.ctx.v[0] = kNearFactor2;
    const int64_t per_wave = centered_targets_per_wave(CP::Ker::ID, sizeof(R) == 8 ? 0 : 1, mode, density);
    const dim3 grid((unsigned)((Nt + per_wave - 1) / per_wave), (unsigned)splits);
    if (mode == 0) launch_centered<CP, R, 0>(a, grid, (int)per_wave, st);
    else if (mode == 1) launch_centered<CP, R, 1>(a, grid, (int)per_wave, st);
    else launch_centered<CP, R, (sizeof(R) == 8 ? 2 : 1)>(a, grid, (int)per_wave, st);
    CENTERED_TRY(hipGetLastError());
    if (splits > 1)
      hipLaunchKernelGGL((reduce_splits_kernel<R>), dim3((unsigned)((Nt * K1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, v_trg, (const R*)partial, Nt * K1, splits, scale);
    return hipGetLastError();
  }
  const int nblk_box = 256;
  const unsigned nb = (unsigned)((Nt + kBlock - 1) / kBlock);
  size_t tmp_bytes = 0;   // size query only: no pointer is dereferenced
  CENTERED_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)Nt, 0,
                                         63, st));
  const size_t n = (size_t)Nt;
  const size_t total = Carver::pad(sizeof(double) * 6 * nblk_box) + 2 * Carver::pad(sizeof(uint64_t) * n) + 2 * Carver::pad(sizeof(uint32_t) * n) +
                       Carver::pad(sizeof(R) * 3 * n) + Carver::pad(sizeof(R) * n * K1) + Carver::pad(tmp_bytes) +
                       (splits > 1 ? Carver::pad(sizeof(R) * (size_t)splits * n * K1) : 0);
  void* base = nullptr;
  CENTERED_TRY(workspace_acquire(st, total, &base));
  Carver cut(base);
  double* part = cut.take<double>(6 * nblk_box);
  uint64_t *keys = cut.take<uint64_t>(n), *keys2 = cut.take<uint64_t>(n);
  uint32_t *idx = cut.take<uint32_t>(n), *idx2 = cut.take<uint32_t>(n);
  R *xts = cut.take<R>(3 * n), *outs = cut.take<R>(n * K1);
  char* tmp = cut.take<char>(tmp_bytes);
  R* partial = (splits > 1) ? cut.take<R>((size_t)splits * n * K1) : nullptr;

  hipLaunchKernelGGL((bbox_partial_kernel<R>), dim3(nblk_box), dim3(kBlock), 0, st, xt, Nt, part);
  hipLaunchKernelGGL((morton_keys_kernel<R>), dim3(nb), dim3(kBlock), 0, st, xt, Nt, (const double*)part, nblk_box, keys, idx);
  CENTERED_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, idx, idx2, (size_t)Nt, 0, 63, st));
  const uint32_t* perm = idx2;
  hipLaunchKernelGGL((gather_points_kernel<R>), dim3(nb), dim3(kBlock), 0, st, xt, perm, Nt, xts);
  CENTERED_TRY(hipMemsetAsync(outs, 0, sizeof(R) * Nt * K1, st));

  EvalArgs<R> a{};
  a.Nt = Nt; a.Ns = Ns; a.xt = xts; a.xs = xs; a.xn = xn; a.f = f; a.v_trg = outs; a.partial = nullptr;
  a.chunk = chunk; a.scale = scale;
  a.ctx.v[0] = kNearFactor2;
  a.partial = partial;
  const int64_t per_wave = centered_targets_per_wave(CP::Ker::ID, sizeof(R) == 8 ? 0 : 1, mode, density);
  const dim3 grid((unsigned)((Nt + per_wave - 1) / per_wave), (unsigned)splits);
  if (mode == 0) launch_centered<CP, R, 0>(a, grid, (int)per_wave, st);
  else if (mode == 1) launch_centered<CP, R, 1>(a, grid, (int)per_wave, st);
  else launch_centered<CP, R, (sizeof(R) == 8 ? 2 : 1)>(a, grid, (int)per_wave, st);
  CENTERED_TRY(hipGetLastError());
  if (splits > 1)
    hipLaunchKernelGGL((reduce_splits_kernel<R>), dim3((unsigned)((Nt * K1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, outs, (const R*)a.partial, Nt * K1, splits, scale);
  hipLaunchKernelGGL((scatter_add_kernel<R>), dim3(nb), dim3(kBlock), 0, st, (const R*)outs, perm, Nt, K1, v_trg);
  return hipGetLastError();
}
// Morton order of n points that are ALREADY on the current device: bbox -> 63-bit keys -> rocPRIM radix sort (stable: ties keep the caller's
// order) -> gather.  d_perm[i] = caller's index of the i-th point of the order, d_sorted = the coordinates in that order.  Temporaries come
// from the stream's scratch block; everything is enqueued on st.  The operator handle (capi.hip: sctl_amd_op_set_targets) uses this instead of
// a host sort: 2^20 points in ~1.5 ms against ~70 ms for std::sort on one host core.
template <class R> hipError_t morton_order_device_t(const R* d_x, int64_t n, R* d_sorted, uint32_t* d_perm, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const int nblk_box = 256;
  const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
  size_t tmp_bytes = 0;
  CENTERED_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 63, st));
  const size_t total = Carver::pad(sizeof(double) * 6 * nblk_box) + 2 * Carver::pad(sizeof(uint64_t) * (size_t)n) + Carver::pad(sizeof(uint32_t) * (size_t)n) + Carver::pad(tmp_bytes);
  void* base = nullptr;
  CENTERED_TRY(workspace_acquire(st, total, &base));
  Carver cut(base);
  double* part = cut.take<double>(6 * nblk_box);
  uint64_t *keys = cut.take<uint64_t>((size_t)n), *keys2 = cut.take<uint64_t>((size_t)n);
  uint32_t* idx = cut.take<uint32_t>((size_t)n);
  char* tmp = cut.take<char>(tmp_bytes);
  hipLaunchKernelGGL((bbox_partial_kernel<R>), dim3(nblk_box), dim3(kBlock), 0, st, d_x, n, part);
  hipLaunchKernelGGL((morton_keys_kernel<R>), dim3(nb), dim3(kBlock), 0, st, d_x, n, (const double*)part, nblk_box, keys, idx);
  CENTERED_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, idx, d_perm, (size_t)n, 0, 63, st));
  hipLaunchKernelGGL((gather_points_kernel<R>), dim3(nb), dim3(kBlock), 0, st, d_x, (const uint32_t*)d_perm, n, d_sorted);
  return hipGetLastError();
}
hipError_t morton_order_device(int real, const void* d_x, int64_t n, void* d_sorted, uint32_t* d_perm, hipStream_t st) {
  return real == 0 /* SCTL_AMD_F64 */ ? morton_order_device_t<double>((const double*)d_x, n, (double*)d_sorted, d_perm, st)
                              : morton_order_device_t<float>((const float*)d_x, n, (float*)d_sorted, d_perm, st);
}

// kernel id -> policy
template <class R>
hipError_t eval_centered(int kernel_id, int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, const R* f, R* v_trg, double scale, int mode, int cus,
                         hipStream_t st, bool presorted, int64_t density) {
  if (kernel_id == Laplace3D_DxU::ID) return eval_centered_t<CenteredDxU<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
  if constexpr (std::is_same<R, double>::value) {   // vector outputs: fp64 (capi.hip: has_centered_path)
    if (kernel_id == Laplace3D_FxdU::ID) return eval_centered_t<CenteredFxdU<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_FxUP::ID) return eval_centered_t<CenteredStokeslet<R, Stokes3D_FxUP>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
  }
  if constexpr (std::is_same<R, float>::value) {    // fp32 kernels with several outputs per target on the matrix cores (mode 0; capi.hip: has_centered_path)
    if (kernel_id == Laplace3D_FxdU::ID) return eval_centered_t<CenteredFxdU<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Laplace3D_FDxUdU::ID) return eval_centered_t<CenteredFusedLaplace<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_FxU::ID) return eval_centered_t<CenteredStokeslet<R, Stokes3D_FxU>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_FSxU::ID) return eval_centered_t<CenteredStokeslet<R, Stokes3D_FSxU>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_FxUP::ID) return eval_centered_t<CenteredStokeslet<R, Stokes3D_FxUP>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_DxU::ID) return eval_centered_t<CenteredStresslet<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
    if (kernel_id == Stokes3D_FxT::ID) return eval_centered_t<CenteredTraction<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
  }
  return eval_centered_t<CenteredFxU<R>, R>(Nt, Ns, xt, xs, xn, f, v_trg, scale, mode, cus, st, presorted, density);
}
template hipError_t eval_centered<double>(int, int64_t, int64_t, const double*, const double*, const double*, const double*, double*, double, int, int, hipStream_t, bool, int64_t);
template hipError_t eval_centered<float>(int, int64_t, int64_t, const float*, const float*, const float*, const float*, float*, double, int, int, hipStream_t, bool, int64_t);

}  // namespace sctl_amd
