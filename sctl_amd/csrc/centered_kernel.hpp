// Laplace single-layer evaluation with tile-centred distances: the fast path of the headline kernel.
//
// Why: the exact kernel (eval_kernel.hpp) is at the fp64 issue limit of its instruction mix — 12 fp64 ops + v_rsq_f64
// per pair; fp64 MFMA shares the fp64 pipe on gfx950 (tools/ubench/mfma_mix.hip), so the only lever
// left is fewer instructions.  Six of the twelve are geometry: d = x_t - x_s (3) and r2 = |d|^2 (3).  With coordinates
// taken relative to a centre c close to the workgroup's targets,
//     r2 = (|x_t'|^2 + |x_s'|^2) - 2 x_t'.x_s'       x' = x - c
// is 1 add + 3 FMAs (|x_t'|^2 and -2 x_t' live in registers, x_s' and |x_s'|^2 in LDS), i.e. four instructions instead
// of six, and it is ACCURATE whenever the source is far from the target cluster: for |x_s'|^2 > 9 Rt^2 (Rt = radius of
// the workgroup's targets about c) the rounding error of r2 is <= 14.5 u relative (u = 2^-53), typically ~4 u — the
// same size as the error of the exact path (d rounded, then three roundings).  Sources that are NOT far
// (|x_s'|^2 <= 9 Rt^2, a few per cent when the targets of a workgroup are spatially compact) take the reference-exact
// path d = x_t - x_s with the r = 0 mask; coincident pairs can only occur there, so the far loop needs no mask at all.
//
// The far/near decision is per (wave, source) — uniform across the wave — and is made when a tile of 64 sources is staged
// into LDS: far and near sources are compacted into two record lists, so both inner loops have uniform trip counts and
// no per-pair branch.  The host side (centered.hip) Morton-sorts the targets first so that the 128 targets of a wave are
// compact; results are scattered back through the permutation.
#pragma once
#include <sctl_amd/device/eval_kernel.hpp>

namespace sctl_amd {

// near  <=>  |x_s - c|^2 <= kNearFactor2 * Rt^2.  A far source is at least (sqrt(kNearFactor2) - 1) Rt from every target of the
// wave, so the cancellation in r^2 = |x_t'|^2 + |x_s'|^2 - 2 x_t'.x_s' amplifies rounding by at most
// (1 + kNearFactor2) / (sqrt(kNearFactor2) - 1)^2 = 5 for the value 4.  Measured on 2^20 x 2^20 uniform points
// (tools/near_factor.py, profiles/r01c_near_factor.txt): 16 -> 473.6 ms, 9 -> 463.5, 4 -> 455.3, 2.25 -> 454.1, with the
// rel-L2 distance to the exact kernel flat at 2.2e-15; 4 takes most of the gain with a bounded amplification.
constexpr double kNearFactor2 = 4.0;

template <class R> __device__ __forceinline__ R wave_min(R v) {
  for (int o = 32; o > 0; o >>= 1) { const R w = __shfl_xor(v, o); v = (w < v) ? w : v; }
  return v;
}
template <class R> __device__ __forceinline__ R wave_max(R v) {
  for (int o = 32; o > 0; o >>= 1) { const R w = __shfl_xor(v, o); v = (w > v) ? w : v; }
  return v;
}

// wave-uniform double -> scalar registers (frees two VGPRs per value in a kernel that is VGPR-limited to 4 waves/SIMD)
__device__ __forceinline__ double uniform_(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ float uniform_(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double sqrt_(double v) { return __builtin_sqrt(v); }
__device__ __forceinline__ float sqrt_(float v) { return __builtin_sqrtf(v); }

// four reals {a, b, c, d} as whole 16-byte LDS words: two double2 or one float4
template <class R> struct Rec4;
template <> struct Rec4<double> {
  typedef double V __attribute__((ext_vector_type(2)));
  static constexpr int NW = 2;
  static __device__ __forceinline__ void put(V* p, double a, double b, double c, double d) { p[0] = V{a, b}; p[1] = V{c, d}; }
  static __device__ __forceinline__ void get(const V* p, double (&o)[4]) { const V u = p[0], w = p[1]; o[0] = u[0]; o[1] = u[1]; o[2] = w[0]; o[3] = w[1]; }
};
template <> struct Rec4<float> {
  typedef float V __attribute__((ext_vector_type(4)));
  static constexpr int NW = 1;
  static __device__ __forceinline__ void put(V* p, float a, float b, float c, float d) { p[0] = V{a, b, c, d}; }
  static __device__ __forceinline__ void get(const V* p, float (&o)[4]) { const V u = p[0]; o[0] = u[0]; o[1] = u[1]; o[2] = u[2]; o[3] = u[3]; }
};

// One wave64 per workgroup: the unit that shares a centre is 64*T Morton-consecutive targets (128 for T = 2), which keeps
// the cluster radius — and with it the fraction of near sources (3.5 % at 2^20 uniform points, vs 9 % for 512 targets) —
// small and evens out the work per SIMD (32 independent workgroups per CU).  Each wave stages its own 64-source tiles;
// the 4x larger L2 read volume (every wave streams all sources: 0.6 TB/s at 2^20) is far below the L2's bandwidth.
constexpr int kWaveBlock = 64;   // lanes per workgroup of the centred kernel
constexpr int kWaveTile = 64;    // sources per LDS tile
#if defined(SCTL_AMD_EXPERIMENTS) && defined(SCTL_AMD_EXP_NEARCAP)   // debugging builds
constexpr int kNearCap = SCTL_AMD_EXP_NEARCAP;   // >= kWaveTile: one tile may bring that many near sources
static_assert(kNearCap >= 64, "a tile of 64 sources must fit the list of pending near sources");
#else
constexpr int kNearCap = 128;    // capacity of the per-wave list of pending near sources
#endif

// fp32 with two targets per lane: the two targets' far pairs as ONE stream of packed instructions.  gfx950 issues v_pk_fma_f32 (two
// FMAs per lane) at the cost of one fp64 FMA, where two v_fma_f32 cost 1.3 (tools/ubench/valu_rates: 2.96 vs 2 x 1.93 cycles per
// instruction per SIMD at 4 waves), and the compiler only packs the final accumulate by itself.  The target-side quantities live as
// {target 0, target 1} register pairs for the whole kernel; a source value is broadcast to both halves by the instruction's op_sel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int P = 1> __device__ __forceinline__ f32x2 rsqrt_pair(f32x2 r2) {   // the P-th power of the result is what the caller accumulates
  f32x2 y = {__builtin_amdgcn_rsqf(r2[0]), __builtin_amdgcn_rsqf(r2[1])};
  if (MODE >= 1) {   // more than 7 digits asked of fp32: the unnormalised Newton step, 2/r (matches rsqrt_newton2; acc_factor carries the 2)
    const f32x2 a = r2 * y;
    y = y * (f32x2{3.0f, 3.0f} - a * y);
    if (MODE == 2) {   // (never launched: fp32 stops at MODE 1, capi.hip mode_for) the factor Ker::acc_factor(2) expects of the P-th power
      const float k = (float)__builtin_pow(rsqrt_pow_factor(2, P), 1.0 / P) / 2;
      y = y * f32x2{k, k};
    }
  }
  return y;
}
// {t0, t1} + the HIGH half of the register pair `zA` in both lanes, as one v_pk_add_f32.  Written out because the compiler, which finds the
// op_sel broadcast for the packed FMAs by itself, copies the high half into a fresh register first for this add (one v_mov_b32 per far
// source = 10 % of the fp32 far loop's issue cycles).  `zA` is the upper half {z', |x_s'|^2} of the record's ds_read_b128.
__device__ __forceinline__ f32x2 pk_add_hi(f32x2 t, f32x2 zA) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(t), "v"(zA));
  return r;
}
// m * (low half of g) + (high half of g) in both lanes: the first step of a dot product whose constant term sits next to its last coefficient
__device__ __forceinline__ f32x2 pk_fma_lo_hi(f32x2 m, f32x2 g) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(m), "v"(g));
  return r;
}

// The target side of a far pair: -2 x_t' and |x_t'|^2 of the lane's T targets.  fp32 with two targets keeps them as {target 0, target 1} register pairs as
// ONLY: built inside the far loop from elements of two arrays, the pairs kept those arrays in scratch memory (32 bytes per lane,
// re-read for every tile).
template <class R, int T> struct FarTargets {
  R m2x[T][3], tt[T];
  __device__ __forceinline__ void set(int j, const R (&p)[3], R t2) {
    tt[j] = t2;
#pragma unroll
    for (int k = 0; k < 3; k++) m2x[j][k] = R(-2) * p[k];
  }
};
template <> struct FarTargets<float, 2> {
  f32x2 mp[3], tp;
  __device__ __forceinline__ void set(int j, const float (&p)[3], float t2) {
    tp[j] = t2;
#pragma unroll
    for (int k = 0; k < 3; k++) mp[k][j] = -2.0f * p[k];
  }
};

// What a kernel needs to run on the centred path: besides {x', y', z', |x_s'|^2} a far source carries XW more reals, written by put_extra /
// put_null and read once per source by load_extra; far_pair is the pair evaluation from the centred quantities into NF far accumulators per
// target.  Near sources go through the kernel's own exact pair (Ker::pack / Ker::pair) into K1 accumulators.  A scalar kernel (NF = K1 = 1) adds
// both into one sum; a kernel whose output needs x_t - x_s itself (a gradient) accumulates MOMENTS over the far sources — with
// x_t - x_s = x_t' - x_s', sum_s A_s (x_t - x_s) = x_t' sum_s A_s - sum_s A_s x_s': no cancellation, |x_t'| <= Rt < |x_s'| / 2 — and finish()
// puts the K1 outputs together once, when the sums leave the registers.  NEAR_CAP: capacity of the per-wave list of pending near sources.
template <class R> struct CenteredFxU {      // u += f / r
  using Ker = Laplace3D_FxU;
  // fp32 keeps the density twice, {f, f}: the packed accumulate then takes it as a register pair as it comes from LDS (the compiler
  // copied the odd ones of four densities read together into fresh registers to broadcast them)
  static constexpr bool DUP = std::is_same<R, float>::value;
  static constexpr int XW = DUP ? 2 : 1, NF = 1, NEAR_CAP = 128;
  template <class RR> static constexpr int targets_per_lane() { return sizeof(RR) == 8 ? 4 : 2; }
  struct Extra { R f, f2; };
  template <int MODE> static __device__ __forceinline__ void put_extra(R* base, int q, const R (&)[3], const R*, const R* f) {
    if (DUP) { base[2 * q] = f[0]; base[2 * q + 1] = f[0]; }
    else base[q] = f[0];
  }
  static __device__ __forceinline__ void put_null(R* base, int q) {
    if (DUP) { base[2 * q] = R(0); base[2 * q + 1] = R(0); }
    else base[q] = R(0);
  }
  static __device__ __forceinline__ Extra load_extra(const R* base, int s) { return DUP ? Extra{base[2 * s], base[2 * s + 1]} : Extra{base[s], R(0)}; }
  static __device__ __forceinline__ void store_extra(R* base, int q, const Extra& e) {
    if (DUP) { base[2 * q] = e.f; base[2 * q + 1] = e.f2; }
    else base[q] = e.f;
  }
  template <int MODE> static __device__ __forceinline__ void far_pair(R& acc, const R (&m2x)[3], R tt, const R (&b)[4], const Extra& e, const RsqConst<R>& K) {
    const R r2 = fma_(m2x[0], b[0], fma_(m2x[1], b[1], fma_(m2x[2], b[2], tt + b[3])));
    acc = fma_(e.f, rsqrt_scaled<MODE, false>(r2, K), acc);   // MODE 1: 2/r, MODE 2: (8/3)/r, as Ker::pair (acc_factor)
  }
  // all T targets of the lane against one far source
  template <int MODE, int T, class KC> static __device__ __forceinline__ void far_pairs(R (&acc)[T][NF], const FarTargets<R, T>& tg, const R (&b)[4], const Extra& e,
                                                                                          const KC& K) {
    if constexpr (std::is_same<R, float>::value && T == 2) {
      f32x2 r2 = pk_add_hi(tg.tp, f32x2{b[2], b[3]});
      r2 = tg.mp[2] * f32x2{b[2], b[2]} + r2;
      r2 = tg.mp[1] * f32x2{b[1], b[1]} + r2;
      r2 = tg.mp[0] * f32x2{b[0], b[0]} + r2;
      const f32x2 a = f32x2{acc[0][0], acc[1][0]} + f32x2{e.f, e.f2} * rsqrt_pair<MODE>(r2);
      acc[0][0] = a[0]; acc[1][0] = a[1];
    } else {
#pragma unroll
      for (int j = 0; j < T; j++) far_pair<MODE>(acc[j][0], tg.m2x[j], tg.tt[j], b, e, K.rsq);
    }
  }
};
template <class R> struct CenteredDxU {      // u += ((x_t - x_s).n f) / r^3, with (x_t - x_s).n f = x_t'.nf - x_s'.nf
  using Ker = Laplace3D_DxU;
  static constexpr int XW = 4, NF = 1, NEAR_CAP = 128;   // XW: {-nf/2 (3 components, to be dotted with m2x = -2 x_t'), -(x_s'.nf)}
  template <class RR> static constexpr int targets_per_lane() { return sizeof(RR) == 8 ? 4 : 2; }
  struct Extra { R g[4]; };
  template <int MODE> static __device__ __forceinline__ void put_extra(R* base, int q, const R (&p)[3], const R* n, const R* f) {
    const R nf[3] = {n[0] * f[0], n[1] * f[0], n[2] * f[0]};
    Rec4<R>::put((typename Rec4<R>::V*)base + q * Rec4<R>::NW, R(-0.5) * nf[0], R(-0.5) * nf[1], R(-0.5) * nf[2], -(p[0] * nf[0] + p[1] * nf[1] + p[2] * nf[2]));
  }
  static __device__ __forceinline__ void put_null(R* base, int q) { Rec4<R>::put((typename Rec4<R>::V*)base + q * Rec4<R>::NW, R(0), R(0), R(0), R(0)); }
  static __device__ __forceinline__ Extra load_extra(const R* base, int s) {
    Extra e;
    Rec4<R>::get((const typename Rec4<R>::V*)base + s * Rec4<R>::NW, e.g);
    return e;
  }
  static __device__ __forceinline__ void store_extra(R* base, int q, const Extra& e) {
    Rec4<R>::put((typename Rec4<R>::V*)base + q * Rec4<R>::NW, e.g[0], e.g[1], e.g[2], e.g[3]);
  }
  template <int MODE> static __device__ __forceinline__ void far_pair(R& acc, const R (&m2x)[3], R tt, const R (&b)[4], const Extra& e, const RsqConst<R>& K) {
    const R r2 = fma_(m2x[0], b[0], fma_(m2x[1], b[1], fma_(m2x[2], b[2], tt + b[3])));
    const R y3 = rsqrt_pow_scaled<MODE, 3, false>(r2, K);        // MODE 1: (2/r)^3, MODE 2: (8/15)/r^3, as Ker::pair (acc_factor)
    const R dn = fma_(m2x[0], e.g[0], fma_(m2x[1], e.g[1], fma_(m2x[2], e.g[2], e.g[3])));
    acc = fma_(dn, y3, acc);
  }
  template <int MODE, int T, class KC> static __device__ __forceinline__ void far_pairs(R (&acc)[T][NF], const FarTargets<R, T>& tg, const R (&b)[4], const Extra& e,
                                                                                          const KC& K) {
    if constexpr (std::is_same<R, float>::value && T == 2) {
      const f32x2 mx = tg.mp[0], my = tg.mp[1], mz = tg.mp[2];
      f32x2 r2 = pk_add_hi(tg.tp, f32x2{b[2], b[3]});
      r2 = mz * f32x2{b[2], b[2]} + r2;
      r2 = my * f32x2{b[1], b[1]} + r2;
      r2 = mx * f32x2{b[0], b[0]} + r2;
      const f32x2 y = rsqrt_pair<MODE, 3>(r2);
      f32x2 dn = pk_fma_lo_hi(mz, f32x2{e.g[2], e.g[3]});
      dn = my * f32x2{e.g[1], e.g[1]} + dn;
      dn = mx * f32x2{e.g[0], e.g[0]} + dn;
      const f32x2 a = f32x2{acc[0][0], acc[1][0]} + dn * (y * y * y);
      acc[0][0] = a[0]; acc[1][0] = a[1];
    } else {
#pragma unroll
      for (int j = 0; j < T; j++) far_pair<MODE>(acc[j][0], tg.m2x[j], tg.tt[j], b, e, K.rsq);
    }
  }
};

// ---- vector outputs: moments over the far sources (fp64; the fp32 forms of these kernels keep the exact path) ----------------------------------------------
// XW reals of a far record as whole 16-byte words
template <class R, int XW> struct ExtraWords {
  static_assert(XW % 4 == 0, "whole Rec4 groups");
  struct Extra { R g[XW]; };
  static __device__ __forceinline__ Extra load_extra(const R* base, int s) {
    Extra e;
#pragma unroll
    for (int w = 0; w < XW / 4; w++) {
      R v[4];
      Rec4<R>::get((const typename Rec4<R>::V*)base + (s * (XW / 4) + w) * Rec4<R>::NW, v);
      e.g[4 * w] = v[0]; e.g[4 * w + 1] = v[1]; e.g[4 * w + 2] = v[2]; e.g[4 * w + 3] = v[3];
    }
    return e;
  }
  static __device__ __forceinline__ void store_extra(R* base, int q, const Extra& e) {
#pragma unroll
    for (int w = 0; w < XW / 4; w++)
      Rec4<R>::put((typename Rec4<R>::V*)base + (q * (XW / 4) + w) * Rec4<R>::NW, e.g[4 * w], e.g[4 * w + 1], e.g[4 * w + 2], e.g[4 * w + 3]);
  }
  static __device__ __forceinline__ void put_null(R* base, int q) {
    Extra e;
#pragma unroll
    for (int k = 0; k < XW; k++) e.g[k] = R(0);
    store_extra(base, q, e);
  }
};
// gradient of the single layer, u_j = sum_s f_s (x_t - x_s)_j / r^3 (kernel_functions.hpp:53-72): with A = f / r^3 the far sources give
// u_j = x_t'_j S0 - S_j, S0 = sum A, S_j = sum A x_s'_j; the record carries {f, f x', f y', f z'}, so a far pair is the 4-instruction distance, the
// reciprocal cube and FOUR accumulations: 13 fp64 instructions + v_rsq_f64 (= 4 more slots) where the exact pair has 15 (3 differences + 3 for r^2, one product
// f / r^3, 3 accumulations; tools/isa_loop_counts.py)
template <class R> struct CenteredFxdU : ExtraWords<R, 4> {
  using Ker = Laplace3D_FxdU;
  using Extra = typename ExtraWords<R, 4>::Extra;
  static constexpr int XW = 4, NF = 4, NEAR_CAP = 128;
  // three targets per lane: a far record is four 16-byte words — with two targets 2 ds_read_b128 per pair, ~8 of the ~18 cycles per CU a wave-pair's arithmetic
  // takes (a wave64 ds_read_b128 costs ~4.3 cycles per CU: tools/ubench/lds_multi_address.hip) — and four targets cost the registers of a third
  // wave per SIMD; against the exact kernel, one box: T = 2 +2.2 .. 2.5 %, T = 3 +4.3 .. 4.6 %, T = 4 +3.2 .. 4.1 % (2^18 and 2^20, profiles/r04_ab_centered_vec.txt)
  template <class RR> static constexpr int targets_per_lane() { return 3; }
  template <int MODE> static __device__ __forceinline__ void put_extra(R* base, int q, const R (&p)[3], const R*, const R* f) {
    Extra e{{f[0], f[0] * p[0], f[0] * p[1], f[0] * p[2]}};
    ExtraWords<R, 4>::store_extra(base, q, e);
  }
  template <int MODE, int T, class KC> static __device__ __forceinline__ void far_pairs(R (&acc)[T][NF], const FarTargets<R, T>& tg, const R (&b)[4], const Extra& e,
                                                                                          const KC& K) {
#pragma unroll
    for (int j = 0; j < T; j++) {
      const R r2 = fma_(tg.m2x[j][0], b[0], fma_(tg.m2x[j][1], b[1], fma_(tg.m2x[j][2], b[2], tg.tt[j] + b[3])));
      const R y3 = rsqrt_pow_scaled<MODE, 3, false>(r2, K.rsq);        // the factor Ker::acc_factor(MODE) accounts for, as Ker::pair
#pragma unroll
      for (int k = 0; k < 4; k++) acc[j][k] = fma_(y3, e.g[k], acc[j][k]);
    }
  }
  // out = near + far: x_t' = -m2x / 2
  template <int MODE> static __device__ __forceinline__ void finish(R (&out)[3], const R (&far)[NF], const R (&near)[3], const R (&m2x)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = near[k] + fma_(R(-0.5) * m2x[k], far[0], -far[1 + k]);
  }
};
// Stokeslet family (kernel_functions.hpp:74-95, 148-198): u_j = sum_s (f_j + r_j (r.f) / r^2) / r, with r = x_t' - x_s'.  With y = C / r, t = C^2 (r.f) / r^2
// and w_j = C^2 f_j - t x_s'_j a far pair adds y w_j to S_j and t y to S_c, and u_j = S_j + x_t'_j S_c: FOUR sums per target where splitting the three terms would
// take seven.  r.f comes as for the double layer, m2x . (-f/2) - x_s'.f, so the record carries {-f/2, -x_s'.f} for the dot product and C^2 f beside it (as the exact
// kernel's record does): 4 (distance) + rsq + 4 (cubic step) + 3 (dot) + 2 (t) + 1 (S_c) + 6 = 20 fp64 instructions + v_rsq_f64 where the exact pair has 21 + v_rsq_f64
// (3 differences and 3 for r^2; its speculative tile loop carries no mask) — ONE instruction of 25 slots' worth, which the near pairs eat: for the Stokeslet itself and for
// Stokes3D_FSxU the path measures -1.4 .. +2.4 % against the exact kernel and is NOT used (profiles/r04_ab_centered_stokeslet.txt; PMC: 21.8 against 22.3 VALU instructions
// per wave-pair, both kernels' vector pipe 90-94 % busy).  S_c is the PRESSURE of Stokes3D_FxUP as it stands, where the exact pair needs two more instructions for it:
// that kernel takes this path, +6.3 % at 2^18, +7.4 % at 2^20.  (Stokes3D_FSxU would add its fourth density to the dot product's constant.)
template <class R, class KER> struct CenteredStokeslet : ExtraWords<R, 8> {
  using Ker = KER;
  using Extra = typename ExtraWords<R, 8>::Extra;
  static constexpr int XW = 8, NF = 4, NEAR_CAP = 64;   // (a near record is six 16-byte words: 128 pending ones would leave two waves per SIMD)
  static_assert(KER::K0 == 3 || KER::K0 == 4, "three force components, optionally a source/sink strength");
  static_assert(KER::K1 == 3 || KER::K1 == 4, "velocity, optionally the pressure");
  template <class RR> static constexpr int targets_per_lane() { return 4; }   // six LDS words per far source: T = 2 / 3 / 4 measured 834 / 826 / 808 ms (Stokeslet, 2^20)
  template <int MODE> static __device__ __forceinline__ void put_extra(R* base, int q, const R (&p)[3], const R*, const R* f) {
    const R c2 = R(rsqrt_scaled_c2(MODE));
    R g3 = -(p[0] * f[0] + p[1] * f[1] + p[2] * f[2]);
    if constexpr (KER::K0 == 4) g3 += f[3];
    Extra e{{R(-0.5) * f[0], R(-0.5) * f[1], R(-0.5) * f[2], g3, c2 * f[0], c2 * f[1], c2 * f[2], R(0)}};
    ExtraWords<R, 8>::store_extra(base, q, e);
  }
  template <int MODE, int T, class KC> static __device__ __forceinline__ void far_pairs(R (&acc)[T][NF], const FarTargets<R, T>& tg, const R (&b)[4], const Extra& e,
                                                                                          const KC& K) {
#pragma unroll
    for (int j = 0; j < T; j++) {
      const R r2 = fma_(tg.m2x[j][0], b[0], fma_(tg.m2x[j][1], b[1], fma_(tg.m2x[j][2], b[2], tg.tt[j] + b[3])));
      const R y = rsqrt_scaled<MODE, false>(r2, K.rsq);                                                              // C / r, as Ker::pair
      const R t = fma_(tg.m2x[j][0], e.g[0], fma_(tg.m2x[j][1], e.g[1], fma_(tg.m2x[j][2], e.g[2], e.g[3]))) * (y * y);   // C^2 (r.f) / r^2
      acc[j][3] = fma_(t, y, acc[j][3]);
#pragma unroll
      for (int k = 0; k < 3; k++) acc[j][k] = fma_(y, fma_(-t, b[k], e.g[4 + k]), acc[j][k]);
    }
  }
  template <int MODE> static __device__ __forceinline__ void finish(R (&out)[KER::K1], const R (&far)[NF], const R (&near)[KER::K1], const R (&m2x)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = near[k] + fma_(R(-0.5) * m2x[k], far[3], far[k]);
    if constexpr (KER::K1 == 4) out[3] = near[3] + far[3];
  }
};
// (The fused single + double layer kernel, potential and gradient — BASELINE config 2 — was written the same way and is NOT kept: u = sum y (q' + w) and
// grad_j = sum y^3 (m'_j + c x_s'_j) - x_t'_j sum y^3 c cost 28 issue slots per far pair where the exact pair has 29 — the two distance slots saved pay for the
// extra moment sum y^3 c — and with the near pairs, a 12-real near record and six LDS words per far source it measured 2.5 .. 3.7 % SLOWER than the exact kernel at
// 2^18 and 2^20: profiles/r04_ab_centered_vec.txt.)

// Which (target tile, source split) a workgroup of the tile-centred kernels takes: see the comment in centered_kernel below (XCD k owns the splits
// [k gridDim.y / 8, (k + 1) gridDim.y / 8) for all tiles).
__device__ __forceinline__ void centered_tile_and_split(unsigned& tile_idx, unsigned& split_idx) {
  tile_idx = blockIdx.x; split_idx = blockIdx.y;
  if ((gridDim.y & 7u) == 0) {
    const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, xcd = b & 7u, j = b >> 3, per = gridDim.y >> 3;
    tile_idx = j % gridDim.x;
    split_idx = xcd * per + j / gridDim.x;
  }
}
// a.xt: Morton-sorted targets; a.v_trg / a.partial: indexed like a.xt (the caller scatters back).
// (asking the compiler for 5-6 waves/SIMD instead of the 4 its registers allow, or unrolling the far loop by 2 or 8 instead of 4, costs 0-3 %:
// profiles/r03_ab_centered_occupancy.txt)
template <class CP, class R, int MODE, int T, int UNR = 4>
__global__ void __launch_bounds__(kWaveBlock) centered_kernel(const EvalArgs<R> a) {
  using V = typename Rec4<R>::V;
  constexpr int NW = Rec4<R>::NW;
  using Ker = typename CP::Ker;
  constexpr int ND = Ker::ND, K0 = Ker::K0, K1 = Ker::K1, NF = CP::NF;
  constexpr bool SCALAR = (K1 == 1 && NF == 1);               // far and near pairs add into ONE sum per target (the Laplace single and double layer)
  constexpr int kNearCap = CP::NEAR_CAP < sctl_amd::kNearCap ? CP::NEAR_CAP : sctl_amd::kNearCap;
  constexpr int NEARW = (Ker::NREC + 3) / 4;                  // Rec4 groups of a near record (the kernel's packed exact record)
  constexpr int XV = (CP::XW * (int)sizeof(R) + 15) / 16;     // 16-byte words of the extra far record
  __shared__ V farB[(kWaveTile + UNR) * NW];                  // {x', y', z', |x_s'|^2}   (+ the leftovers of earlier tiles)
  __shared__ V farXv[(kWaveTile + UNR) * (XV > 0 ? XV : 1)];    // the policy's extra far reals (density, or the normal terms)
  __shared__ V nearA[(kNearCap + 2) * NW * NEARW];            // packed exact records; near sources are collected over
                                                              // several tiles and evaluated in batches, so the exact loop runs
                                                              // rarely and with a long trip count
  R* const farX = (R*)farXv;

  const int lane = threadIdx.x;
  // Which (target tile, source split) this workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs in launch order
  // (b = y * gridDim.x + x; blocks b, b + 8, ... share an XCD and its L2).  With the plain (x, y) = blockIdx every XCD sees every
  // source split — 8 x the source data through the fabric —; when the splits divide by 8, XCD k instead owns the splits
  // [k * gridDim.y / 8, (k + 1) * gridDim.y / 8) for ALL tiles: one split (2 MB at 2^20 sources / 16) stays in its 4 MB L2 while the
  // tiles stream by.  Same work per XCD; the results do not depend on the mapping.
  unsigned tile_idx, split_idx;
  centered_tile_and_split(tile_idx, split_idx);   // (A/B against the plain mapping: profiles/r02_ab_xcd_map.txt)
  const int64_t tbase = (int64_t)tile_idx * (kWaveBlock * T);
  const typename Ker::template Consts<R> K(nullptr);

  // ---- targets of this lane, cluster centre (bounding-box midpoint) and radius --------------------------------
  R xt[T][3];
  R c[3];
  {
    R lo[3] = {max_finite<R>(), max_finite<R>(), max_finite<R>()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
#pragma unroll
    for (int j = 0; j < T; j++) {
      int64_t t = tbase + j * kWaveBlock + lane;
      if (t >= a.Nt) t = a.Nt - 1;   // tail lanes repeat the last target; never stored
#pragma unroll
      for (int k = 0; k < 3; k++) {
        xt[j][k] = a.xt[t * 3 + k];
        lo[k] = (xt[j][k] < lo[k]) ? xt[j][k] : lo[k];
        hi[k] = (xt[j][k] > hi[k]) ? xt[j][k] : hi[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = uniform_(R(0.5) * wave_min(lo[k]) + R(0.5) * wave_max(hi[k]));
  }
  FarTargets<R, T> tg;
  R rt2 = 0;
#pragma unroll
  for (int j = 0; j < T; j++) {
    const R p[3] = {xt[j][0] - c[0], xt[j][1] - c[1], xt[j][2] - c[2]};
    const R t2 = len2(p);
    rt2 = (t2 > rt2) ? t2 : rt2;
    tg.set(j, p, t2);
  }
  rt2 = uniform_(wave_max(rt2));
  const R near_r2 = R(a.ctx.v[0]) * rt2;   // ctx.v[0] = kNearFactor2; NaN coordinates fail every comparison => "near" => exact path

  R acc[T][K1];                      // the exact (near) pairs' sums; a scalar kernel's far sums too
  R facc[T][SCALAR ? 1 : NF];        // the far pairs' moments (vector outputs)
#pragma unroll
  for (int j = 0; j < T; j++) {
#pragma unroll
    for (int k = 0; k < K1; k++) acc[j][k] = 0;
#pragma unroll
    for (int k = 0; k < (SCALAR ? 1 : NF); k++) facc[j][k] = 0;
  }

  const int64_t s_begin = (int64_t)split_idx * a.chunk;
  const int64_t s_end = (s_begin + a.chunk < a.Ns) ? s_begin + a.chunk : a.Ns;
  const int64_t len = (s_end > s_begin) ? s_end - s_begin : 0;
  const int ntile = (int)((len + kWaveTile - 1) / kWaveTile);

  // software pipeline: the next tile's source is loaded into registers while the current tile is evaluated
  R x[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, f[K0];
#pragma unroll
  for (int k = 0; k < K0; k++) f[k] = 0;
  auto load_source = [&](int it) {
    const int64_t s = s_begin + (int64_t)it * kWaveTile + lane;
    if (s < s_end) {
#pragma unroll
      for (int k = 0; k < 3; k++) x[k] = a.xs[s * 3 + k];
#pragma unroll
      for (int k = 0; k < ND; k++) nrm[k] = a.xn[s * ND + k];
#pragma unroll
      for (int k = 0; k < K0; k++) f[k] = a.f[s * K0 + k];
    }
  };
  if (ntile > 0) load_source(0);

  // ---- near sources: the reference-exact pair (d = x_t - x_s, masked at r = 0), evaluated in batches -----------
  const R far_off = R(1.0e3) * (R(1) + sqrt_(rt2));
  auto put_near = [&](int q, const R (&xq)[3], const R (&nq)[3], const R (&fq)[K0]) {
    R rec[4 * NEARW] = {};
    pack_record<Ker, R, MODE>(rec, xq, nq, fq);
#pragma unroll
    for (int g = 0; g < NEARW; g++) Rec4<R>::put(nearA + (q * NEARW + g) * NW, rec[4 * g], rec[4 * g + 1], rec[4 * g + 2], rec[4 * g + 3]);
  };
  int nn = 0;   // pending near sources in nearA (wave-uniform)
  auto flush_near = [&]() {
    if (nn & 1) {   // pad to an even count with a null source
      if (lane == 0) {
        const R xq[3] = {c[0] + far_off, c[1], c[2]}, nq[3] = {0, 0, 0}, fq[K0] = {};
        put_near(nn, xq, nq, fq);
      }
      __syncthreads();
    }
    R xo[T][3];   // the original target coordinates are needed only here: reloaded (L2 hit) rather than kept in 12 VGPRs
#pragma unroll
    for (int j = 0; j < T; j++) {
      int64_t t = tbase + j * kWaveBlock + lane;
      if (t >= a.Nt) t = a.Nt - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) xo[j][k] = a.xt[t * 3 + k];
    }
    for (int s = 0; s < nn; s += 2) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        R q[4 * NEARW];
#pragma unroll
        for (int g = 0; g < NEARW; g++) {
          R w[4];
          Rec4<R>::get(nearA + ((s + u) * NEARW + g) * NW, w);
          q[4 * g] = w[0]; q[4 * g + 1] = w[1]; q[4 * g + 2] = w[2]; q[4 * g + 3] = w[3];
        }
#pragma unroll
        for (int j = 0; j < T; j++) {
          const R d[3] = {xo[j][0] - q[0], xo[j][1] - q[1], xo[j][2] - q[2]};
          Ker::template pair<R, MODE, true>(acc[j], d, q, a.ctx, K);
          // One pair after the other.  Round 3 met near sums that differed from RUN TO RUN in one instantiation (the matrix-core double-layer kernel with 128 targets
          // per wave, centered_mfma_kernel.hpp): whole contributions lost in lanes 48-63.  Round 4 pinned it down on that kernel's assembly
          // (profiles/r04_near_fault_report.md): it needs the PACKED-fp32 and transcendental instructions the compiler interleaves in this loop — the SLP
          // vectoriser packs the lane's targets — to meet another wave's bursts of v_rsq_f32 in the far loop; idle instructions cure it only behind every one
          // of them, and an overwritten transcendental source (the first suspect) has nothing to do with it.  Two things keep it out: this unit is compiled
          // with -fno-slp-vectorize (Makefile: no packed instruction in this loop; tools/check_isa_rules.py holds the assembly to it), which alone gives clean
          // runs, and this fence, round 3's cure by experiment, which costs nothing.
#if !(defined(SCTL_AMD_EXPERIMENTS) && defined(SCTL_AMD_EXP_NO_NEAR_FENCE))   // (the A/B build of tools/near_determinism.py)
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
    }
    nn = 0;
  };

  // One tile: classify each source as far / near, compact the two lists (far records behind the `carry` left over from earlier tiles), start
  // the loads of the next tile.  Returns the number of far sources of this tile.
  auto stage_tile = [&](int it, int carry) -> int {
    const int ns = (it == ntile - 1) ? (int)(len - (int64_t)it * kWaveTile) : kWaveTile;
    const bool valid = lane < ns;
    const R p[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
    const R ss = len2(p);
    const bool is_far = valid && (ss > near_r2);
    const bool is_near = valid && !is_far;
    const unsigned long long bf = __ballot(is_far), bn = __ballot(is_near);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nfar = __popcll(bf), nnear = __popcll(bn);
    __syncthreads();   // previous tile's far records fully consumed
    if (nn + nnear > kNearCap) {   // wave-uniform: evaluate the pending near sources before the list overflows
      flush_near();
      __syncthreads();
    }
    if (is_far) {
      const int q = carry + __popcll(bf & below);
      Rec4<R>::put(farB + q * NW, p[0], p[1], p[2], ss);
      CP::template put_extra<MODE>(farX, q, p, nrm, f);
    } else if (is_near) {
      put_near(nn + __popcll(bn & below), x, nrm, f);
    }
    nn += nnear;
    return nfar;
  };
  // records [0, m) of the far list, m a multiple of UNR: the 4-instruction distance, no mask.  Per-call partial sums, folded into acc at the
  // end (two-level summation: matters for fp32 at Ns = 2^23)
  auto run_far = [&](int m) {
    R tacc[T][NF];
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int k = 0; k < NF; k++) tacc[j][k] = 0;
    for (int s = 0; s < m; s += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        R b[4];
        Rec4<R>::get(farB + (s + u) * NW, b);
        const typename CP::Extra e = CP::load_extra(farX, s + u);
        CP::template far_pairs<MODE, T>(tacc, tg, b, e, K);
      }
    }
#pragma unroll
    for (int j = 0; j < T; j++) {
      if constexpr (SCALAR) acc[j][0] += tacc[j][0];
      else {
#pragma unroll
        for (int k = 0; k < NF; k++) facc[j][k] += tacc[j][k];
      }
    }
  };
  auto put_null_far = [&](int q) { Rec4<R>::put(farB + q * NW, far_off, R(0), R(0), far_off * far_off); CP::put_null(farX, q); };   // zero density at ~1e3 cluster radii: contributes exactly 0

  // The far list is consumed UNR records at a time.  fp64: what is left over (< UNR) moves to the front of the list for the next tile instead
  // of being padded with null sources (1.5 evaluations in ~62 per tile): +0.9 % (A/B, profiles/r02_ab_far_carry.txt).  fp32, whose tile
  // costs a third of the cycles, gains nothing from it and pads every tile.
  if constexpr (sizeof(R) == 8) {
    int carry = 0;   // far records left over from the previous tiles (wave-uniform, < UNR)
    for (int it = 0; it < ntile; it++) {
      const int n = carry + stage_tile(it, carry), m = n & ~(UNR - 1);
      if (it + 1 < ntile) load_source(it + 1);
      __syncthreads();
      run_far(m);
      carry = n - m;
      if (m > 0 && lane < carry) {   // the leftovers to the front (one wave: its LDS operations complete in program order)
        R b[4];
        Rec4<R>::get(farB + (m + lane) * NW, b);
        const typename CP::Extra e = CP::load_extra(farX, m + lane);
        Rec4<R>::put(farB + lane * NW, b[0], b[1], b[2], b[3]);
        CP::store_extra(farX, lane, e);
      }
    }
    __syncthreads();
    if (carry > 0) {   // the last leftovers, padded once
      if (lane < UNR - carry) put_null_far(carry + lane);
      __syncthreads();
      run_far(UNR);
    }
  } else {
    for (int it = 0; it < ntile; it++) {
      const int nfar = stage_tile(it, 0), padded = (nfar + UNR - 1) & ~(UNR - 1);
      if (lane < UNR - 1 && nfar + lane < padded) put_null_far(nfar + lane);
      if (it + 1 < ntile) load_source(it + 1);
      __syncthreads();
      run_far(padded);
    }
  }
  __syncthreads();
  flush_near();

#pragma unroll
  for (int j = 0; j < T; j++) {
    const int64_t t = tbase + j * kWaveBlock + lane;
    if constexpr (SCALAR) {
      if (t < a.Nt) {
        if (gridDim.y == 1) a.v_trg[t] += acc[j][0] * a.scale;
        else a.partial[(int64_t)split_idx * a.Nt + t] = acc[j][0];
      }
    } else {
      R out[K1];
      CP::template finish<MODE>(out, facc[j], acc[j], tg.m2x[j]);
      finish_acc<Ker, R, MODE>(out);          // (what the kernel itself leaves for the end, e.g. the fused kernel's factor on its potential)
      if (t < a.Nt) {
#pragma unroll
        for (int k = 0; k < K1; k++) {
          if (gridDim.y == 1) a.v_trg[t * K1 + k] += out[k] * a.scale;
          else a.partial[((int64_t)split_idx * a.Nt + t) * K1 + k] = out[k];
        }
      }
    }
  }
}

// ---- Morton ordering of the targets (host side drives these through rocPRIM's radix sort, capi.hip) ---------
// per-block bounding boxes -> bbox[block][6]
template <class R> __global__ void __launch_bounds__(kBlock) bbox_partial_kernel(const R* x, int64_t n, double* part) {
  __shared__ double red[4 * 6];
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308}, hi[3] = {-lo[0], -lo[0], -lo[0]};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double v = (double)x[i * 3 + k];
      lo[k] = (v < lo[k]) ? v : lo[k];
      hi[k] = (v > hi[k]) ? v : hi[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = wave_max(hi[k]); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { red[wave * 6 + k] = lo[k]; red[wave * 6 + 3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    double v = red[k];
    for (int w = 1; w < 4; w++) v = (k < 3) ? ((red[w * 6 + k] < v) ? red[w * 6 + k] : v) : ((red[w * 6 + k] > v) ? red[w * 6 + k] : v);
    part[blockIdx.x * 6 + k] = v;
  }
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// final bbox from the per-block ones (read by every thread: nblk is small), then 63-bit Morton keys + identity index
template <class R> __global__ void __launch_bounds__(kBlock) morton_keys_kernel(const R* x, int64_t n, const double* part, int nblk, uint64_t* keys, uint32_t* idx) {
  __shared__ double box[6];
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    double v = part[k];
    for (int b = 1; b < nblk; b++) v = (k < 3) ? ((part[b * 6 + k] < v) ? part[b * 6 + k] : v) : ((part[b * 6 + k] > v) ? part[b * 6 + k] : v);
    box[k] = v;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double w = box[3 + k] - box[k];
    const double q = (w > 0) ? ((double)x[i * 3 + k] - box[k]) / w * 2097151.0 : 0.0;
    const uint64_t qi = (q > 0) ? (uint64_t)((q < 2097151.0) ? q : 2097151.0) : 0ull;   // NaN -> 0
    key |= spread21(qi) << k;
  }
  keys[i] = key;
  idx[i] = (uint32_t)i;
}

template <class R> __global__ void __launch_bounds__(kBlock) gather_points_kernel(const R* x, const uint32_t* perm, int64_t n, R* out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t s = perm[i];
#pragma unroll
  for (int k = 0; k < 3; k++) out[i * 3 + k] = x[s * 3 + k];
}

// v_trg[perm[i]*K1 + k] += sorted[i*K1 + k]   (sorted already carries the scale factor)
template <class R> __global__ void __launch_bounds__(kBlock) scatter_add_kernel(const R* sorted, const uint32_t* perm, int64_t n, int k1, R* v_trg) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t t = perm[i];
  for (int k = 0; k < k1; k++) v_trg[t * k1 + k] += sorted[i * k1 + k];
}

}  // namespace sctl_amd
