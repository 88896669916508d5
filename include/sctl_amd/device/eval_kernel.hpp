// All-pairs evaluation kernels for gfx950 (MI355X): the device replacement of the loop nest in
// GenericKernel<uKernel>::Eval (reference include/sctl/generic-kernel.txx:76-189).
//
// Work decomposition (nothing like the reference's "OpenMP over target SIMD blocks, serial over sources"):
//   * a workgroup is 256 lanes = 4 wave64; each lane owns T targets in registers (coords + K1 accumulators);
//   * grid.x tiles the targets (256*T per workgroup), grid.y splits the SOURCE range so that small target
//     counts still fill 256 CUs; each split writes unscaled partial sums, a second kernel adds them in a fixed
//     order (deterministic: no atomics) — with one split the main kernel accumulates into v_trg directly;
//   * sources stream through LDS in tiles of 256 packed records (ukernels.hpp: pack()); a record is read back
//     with ds_read_b128 at one address for the whole wave (LDS broadcast, no bank conflicts) and reused for
//     the T targets of the lane.
// The kernel is fp64/fp32 VALU-bound by five orders of magnitude over its HBM traffic,
// so there is no double buffering of the tile: the tile fill is < 1 % of the tile's compute and other
// resident workgroups cover it.
#pragma once
#include <type_traits>

#include "ukernels.hpp"

namespace sctl_amd {

constexpr int kBlock = 256;   // lanes per workgroup
constexpr int kTile = 256;    // sources per LDS tile (one per lane at fill time)

template <class R> struct EvalArgs {
  int64_t Nt, Ns;
  const R* xt;      // [Nt*3]
  const R* xs;      // [Ns*3]
  const R* xn;      // [Ns*ND] or null
  const R* f;       // [Ns*K0]
  R* v_trg;         // [Nt*K1], accumulated into (only touched by the main kernel when gridDim.y == 1)
  R* partial;       // [gridDim.y][Nt*K1] unscaled partial sums when gridDim.y > 1
  int64_t chunk;    // sources per split, a multiple of kTile
  R scale;
  KerCtx ctx;
};

template <class R> struct VecOf;
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };

// sources per unrolled loop body: 4-8 independent accumulation chains in flight per lane (other unroll factors and forced occupancies were timed in rounds 1-3:
// profiles/r03_ab_helmholtz_unroll.txt, r03_ab_centered_occupancy.txt)
template <int T, int K1> struct UnrollOf { static constexpr int value = (T * K1 >= 8) ? 1 : ((T * K1 >= 3) ? 2 : 4); };

template <class R> __device__ __forceinline__ R max_finite();
template <> __device__ __forceinline__ double max_finite<double>() { return 1.7976931348623157e308; }
template <> __device__ __forceinline__ float max_finite<float>() { return 3.402823466e38f; }
__device__ __forceinline__ double fabs_(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float fabs_(float x) { return __builtin_fabsf(x); }

template <class Ker, class R, int MODE, int T>
__global__ void __launch_bounds__(kBlock) eval_kernel(const EvalArgs<R> a) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NREC = Ker::NREC;
  using V = typename VecOf<R>::type;
  constexpr int VN = VecOf<R>::N;
  constexpr int NV = (NREC + VN - 1) / VN;     // 16-byte LDS words per record
  constexpr int NRECP = NV * VN;               // record padded to whole words (fp32: multiples of 4 reals)
  __shared__ V tile[kTile * NV];

  const int tid = threadIdx.x;
  // (tile, split) of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), so with the plain
  // mapping every XCD's L2 streams every source split.  When the splits come in multiples of 8, XCD k — the workgroups b = k mod 8 —
  // takes the splits [k S/8, (k+1) S/8), ONE AT A TIME: all target tiles against its first split, then all against the next.  A split
  // (<= 2 MB by the planner's rule) then lives in one XCD's 4 MB L2 while that XCD's workgroups go through the tiles.  (Tile-major
  // order inside an XCD keeps S/8 splits hot at once: 4 x 2 MB measured 87 % L2 hits on the sources where this order has them all.)
  unsigned tile_x = blockIdx.x, split_y = blockIdx.y;
  if (gridDim.y >= 8 && (gridDim.y & 7) == 0) {
    const unsigned b = blockIdx.x + gridDim.x * blockIdx.y, i = b >> 3;
    tile_x = i % gridDim.x;
    split_y = (b & 7) * (gridDim.y >> 3) + i / gridDim.x;
  }
  const int64_t tbase = (int64_t)tile_x * (kBlock * T);
  using KC = typename Ker::template Consts<R>;
  constexpr int SCRATCH = AllPairsScratch<KC>::value;   // this evaluator's workgroups are long-lived: a kernel may ask for larger tables here
  __shared__ double kscratch[SCRATCH > 0 ? SCRATCH : 1];
  const KC K = make_consts<KC>(kscratch, SCRATCH, a.ctx, MODE);

  R xt[T][3], acc[T][K1];
#pragma unroll
  for (int j = 0; j < T; j++) {
    int64_t t = tbase + j * kBlock + tid;
    if (t >= a.Nt) t = a.Nt - 1;   // tail lanes recompute the last target; never stored
#pragma unroll
    for (int k = 0; k < 3; k++) xt[j][k] = a.xt[t * 3 + k];
#pragma unroll
    for (int k = 0; k < K1; k++) acc[j][k] = 0;
  }

  const int64_t s_begin = (int64_t)split_y * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  const int64_t len = (s_end > s_begin) ? s_end - s_begin : 0;
  const int ntile = (int)((len + kTile - 1) / kTile);
  bool always_masked = (ntile < 4);   // few tiles: speculation cannot pay for a repair
  int repairs = 0;

  // One target per lane is what small target counts run (make_plan): few tiles per workgroup, every workgroup of the chip in step,
  // so nobody covers a tile fill's HBM latency.  Those instantiations fetch the NEXT tile's source into registers while the current
  // tile is evaluated (155 -> 150 us at 2^14 x 2^14); with more targets per lane the registers are worth more as occupancy.
  constexpr bool PREFETCH = (T == 1);
  R px[3] = {0, 0, 0}, pn[3] = {0, 0, 0}, pf[K0];
#pragma unroll
  for (int k = 0; k < K0; k++) pf[k] = 0;
  auto fetch_source = [&](int it) {
    const int64_t s = s_begin + (int64_t)it * kTile + tid;
    if (s < s_end) {
#pragma unroll
      for (int k = 0; k < 3; k++) px[k] = a.xs[s * 3 + k];
#pragma unroll
      for (int k = 0; k < ND; k++) pn[k] = a.xn[s * ND + k];
#pragma unroll
      for (int k = 0; k < K0; k++) pf[k] = a.f[s * K0 + k];
    }
  };
  if (PREFETCH && ntile > 0) fetch_source(0);

  for (int it = 0; it < ntile; it++) {
    const int ns = (it == ntile - 1) ? (int)(len - (int64_t)it * kTile) : kTile;   // wave-uniform
    __syncthreads();   // previous tile fully consumed
    if (!PREFETCH) fetch_source(it);
    if (tid < ns) {
      R rec[NRECP] = {};
      pack_record<Ker, R, MODE>(rec, px, pn, pf);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        V w;
#pragma unroll
        for (int e = 0; e < VN; e++) w[e] = rec[v * VN + e];
        tile[tid * NV + v] = w;
      }
    }
    if (PREFETCH && it + 1 < ntile) fetch_source(it + 1);   // in flight during this tile's arithmetic
    __syncthreads();

    // Per-tile accumulators (folded into the running sums once per tile: two-level summation, which also bounds
    // fp32 error growth at Ns = 2^23).  The tile is first evaluated WITHOUT the r = 0 mask; a
    // coincident pair poisons the tile sums with inf/NaN, in which case the whole wave re-runs this tile masked.
    // A wave that had to repair more than 1/8 of its tiles (e.g. targets == sources in shuffled order at small N)
    // stops speculating and runs masked from then on.
    R tacc[T][K1];
    auto run_tile_v = [&](auto masked_tag, auto variant_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      constexpr int VARIANT = decltype(variant_tag)::value;
      K.begin_tile();
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) tacc[j][k] = 0;
      // one source against the T targets of this lane; the record is read at ONE LDS address by the whole wave
      // (broadcast) and kept in registers for the T pair evaluations
      auto one_source = [&](int s) {
        R rec[NRECP];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const V w = tile[s * NV + v];
#pragma unroll
          for (int e = 0; e < VN; e++) rec[v * VN + e] = w[e];
        }
#pragma unroll
        for (int j = 0; j < T; j++) {
          const R d[3] = {xt[j][0] - rec[0], xt[j][1] - rec[1], xt[j][2] - rec[2]};
          if constexpr (KC::HAS_VARIANT) Ker::template pair<R, MODE, MASKED, VARIANT>(tacc[j], d, rec, a.ctx, K);
          else Ker::template pair<R, MODE, MASKED>(tacc[j], d, rec, a.ctx, K);
        }
      };
      if (ns == kTile) {   // every tile but possibly the last: constant trip count, unrolled
#pragma unroll UnrollOf<T, Ker::K1>::value
        for (int s = 0; s < kTile; s++) one_source(s);
      } else {
        for (int s = 0; s < ns; s++) one_source(s);
      }
    };
    // a kernel with a launch-uniform special case (Helmholtz: real wavenumber) gets its own straight-line copy of the loop
    auto run_tile = [&](auto masked_tag) {
      if constexpr (KC::HAS_VARIANT) {
        const int v = (int)K.variant(a.ctx);
        if constexpr (NumVariants<KC>::value > 2) {
          if (v == 3) run_tile_v(masked_tag, std::integral_constant<int, 3>());
          else if (v == 2) run_tile_v(masked_tag, std::integral_constant<int, 2>());
          else if (v == 1) run_tile_v(masked_tag, std::integral_constant<int, 1>());
          else run_tile_v(masked_tag, std::integral_constant<int, 0>());
        } else {
          if (v) run_tile_v(masked_tag, std::integral_constant<int, 1>());
          else run_tile_v(masked_tag, std::integral_constant<int, 0>());
        }
      } else {
        run_tile_v(masked_tag, std::integral_constant<int, 0>());
      }
    };
    bool repaired = true;
    if (!always_masked) {
      run_tile(std::false_type());
      bool bad = K.tile_bad(a.ctx);
#pragma unroll
      for (int j = 0; j < T; j++)
#pragma unroll
        for (int k = 0; k < K1; k++) bad |= !(fabs_(tacc[j][k]) <= max_finite<R>());
      repaired = __any(bad);                 // wave-uniform
      if (repaired && (++repairs) * 8 > ntile) always_masked = true;
    }
    if (repaired) run_tile(std::true_type());
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int k = 0; k < K1; k++) acc[j][k] += tacc[j][k];
  }

#pragma unroll
  for (int j = 0; j < T; j++) {
    const int64_t t = tbase + j * kBlock + tid;
    finish_acc<Ker, R, MODE>(acc[j]);
    if (t < a.Nt) {
      if (gridDim.y == 1) {
#pragma unroll
        for (int k = 0; k < K1; k++) a.v_trg[t * K1 + k] += acc[j][k] * a.scale;   // generic-kernel.txx:184
      } else {
        R* p = a.partial + ((int64_t)split_y * a.Nt + t) * K1;
#pragma unroll
        for (int k = 0; k < K1; k++) p[k] = acc[j][k];
      }
    }
  }
}

// v_trg[i] += scale * sum_y partial[y][i], y in increasing order (deterministic).
template <class R> __global__ void __launch_bounds__(kBlock) reduce_splits_kernel(R* v_trg, const R* partial, int64_t n, int splits, R scale) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  R s = 0;
  for (int y = 0; y < splits; y++) s += partial[(int64_t)y * n + i];
  v_trg[i] += s * scale;
}

// Dense operator block, GenericKernel::KernelMatrix (generic-kernel.txx:191-307):
//   M[(s*K0 + k0)][(t*K1 + k1)] = scale * U(x_t - x_s, n_s)[k0][k1],   M is (Ns*K0) x (Nt*K1), overwritten.
// One lane per (s, t) pair, t fastest so that stores of one k0,k1 plane are K1-strided runs; the matrix is
// output-bandwidth bound (K0*K1*sizeof(R) bytes per pair), so no LDS staging of inputs is needed.
// U[k0][k1] is obtained by feeding unit densities through the same pair() routine the evaluator uses.
template <class Ker, class R, int MODE>
__global__ void __launch_bounds__(kBlock) matrix_kernel(int64_t Nt, int64_t Ns, const R* xt, const R* xs, const R* xn, R* M, R scale, KerCtx ctx) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NREC = Ker::NREC;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  using KC = typename Ker::template Consts<R>;
  __shared__ double kscratch[KC::LDS_DOUBLES > 0 ? KC::LDS_DOUBLES : 1];
  const KC K = make_consts<KC>(kscratch, ctx, MODE);        // before the early return: the constructor may synchronise the workgroup
  if (t >= Nt) return;
  for (int64_t s = blockIdx.y; s < Ns; s += gridDim.y) {   // gridDim.y is capped at 65535
  R x[3], n[3] = {0, 0, 0}, d[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { x[k] = xs[s * 3 + k]; d[k] = xt[t * 3 + k] - x[k]; }
#pragma unroll
  for (int k = 0; k < ND; k++) n[k] = xn[s * ND + k];
#pragma unroll
  for (int k0 = 0; k0 < K0; k0++) {
    R f[K0], rec[NREC], acc[K1];
#pragma unroll
    for (int k = 0; k < K0; k++) f[k] = (k == k0) ? R(1) : R(0);
#pragma unroll
    for (int k = 0; k < K1; k++) acc[k] = 0;
    pack_record<Ker, R, MODE>(rec, x, n, f);
    Ker::template pair<R, MODE, true>(acc, d, rec, ctx, K);
    finish_acc<Ker, R, MODE>(acc);
    R* row = M + ((s * K0 + k0) * Nt + t) * K1;
#pragma unroll
    for (int k = 0; k < K1; k++) row[k] = acc[k] * scale;
  }
  }
}

}  // namespace sctl_amd

namespace sctl_amd {

// Many small operator blocks in ONE launch (BoundaryIntegralOp::SetupNear builds one block per element,
// boundary_integral.txx:946-1009): a workgroup takes a tile of 64 targets of one block, its four waves interleave the
// sources.  Block b is stored like a single KernelMatrix result, (Ns_b*K0) x (Nt_b*K1) row-major, at entry offset m_off.
struct MatTile {
  int64_t t_off, s_off, m_off;   // first target / first source of the block in the concatenated coordinate arrays; first entry of the block
  int32_t nt, ns, t0, pad;
};
template <class Ker, class R, int MODE>
__global__ void __launch_bounds__(kBlock) matrix_batch_kernel(const MatTile* __restrict__ tiles, const R* xt, const R* xs, const R* xn, R* M, R scale, KerCtx ctx) {
  constexpr int K0 = Ker::K0, K1 = Ker::K1, ND = Ker::ND, NREC = Ker::NREC;
  using KC = typename Ker::template Consts<R>;
  __shared__ double kscratch[KC::LDS_DOUBLES > 0 ? KC::LDS_DOUBLES : 1];
  const KC K = make_consts<KC>(kscratch, ctx, MODE);
  const MatTile w = tiles[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = w.t0 + lane;
  if (t >= w.nt) return;
  R x_t[3];
#pragma unroll
  for (int k = 0; k < 3; k++) x_t[k] = xt[(w.t_off + t) * 3 + k];
  for (int s = wave; s < w.ns; s += kBlock / 64) {
    R x[3], n[3] = {0, 0, 0}, d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { x[k] = xs[(w.s_off + s) * 3 + k]; d[k] = x_t[k] - x[k]; }
#pragma unroll
    for (int k = 0; k < ND; k++) n[k] = xn[(w.s_off + s) * ND + k];
#pragma unroll
    for (int k0 = 0; k0 < K0; k0++) {
      R f[K0], rec[NREC], acc[K1];
#pragma unroll
      for (int k = 0; k < K0; k++) f[k] = (k == k0) ? R(1) : R(0);
#pragma unroll
      for (int k = 0; k < K1; k++) acc[k] = 0;
      pack_record<Ker, R, MODE>(rec, x, n, f);
      Ker::template pair<R, MODE, true>(acc, d, rec, ctx, K);
      finish_acc<Ker, R, MODE>(acc);
      R* row = M + w.m_off + (((int64_t)s * K0 + k0) * w.nt + t) * K1;
#pragma unroll
      for (int k = 0; k < K1; k++) row[k] = acc[k] * scale;
    }
  }
}

}  // namespace sctl_amd

namespace sctl_amd {

// Far-field pre/post steps of BoundaryIntegralOp::ComputeFarField on the device:
//   f[s*K0 + k] *= w[s]                                      the quadrature weights     boundary_integral.txx:1040-1052
//   u[t*K1_ + k] = sum_l v[(t*K1_ + k)*3 + l] * n[t*3 + l]     dot with target normals    boundary_integral.txx:1060-1071
template <class R> __global__ void __launch_bounds__(kBlock) scale_density_kernel(R* f, const R* w, int64_t n, int k0) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n * k0) f[i] *= w[i / k0];
}
template <class R> __global__ void __launch_bounds__(kBlock) normal_dot_kernel(const R* v, const R* nrm, R* u, int64_t nt, int k1r) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nt * k1r) return;
  const int64_t t = i / k1r;
  u[i] = v[i * 3] * nrm[t * 3] + v[i * 3 + 1] * nrm[t * 3 + 1] + v[i * 3 + 2] * nrm[t * 3 + 2];
}

}  // namespace sctl_amd
