// Device micro-kernels for gfx950: one struct per Green's function.
//
// Each struct replaces one functor of the reference's include/sctl/kernel_functions.hpp (cited per struct)
// but is NOT a transcription: the reference builds the K0 x K1 matrix U and then does K0*K1 FMAs
// (generic-kernel.txx:81-90); here every kernel is the contracted form  acc[k1] += sum_k0 U[k0][k1] f[k0]
// written with the fewest fp64 VALU instructions (MI355X fp64 FMA issues at 4 cycles per wave64, the
// reciprocal-square-root seed at 16 — tools/ubench/valu_rates.hip), and source records are pre-packed in
// LDS (e.g. normal*density for double-layer kernels) so that the per-pair work is minimal.
//
// Interface used by eval_kernel.hpp:
//   K0, K1, ND          SrcDim, TrgDim, NormalDim (generic-kernel.hpp:59-84)
//   NREC                reals per packed source record in LDS (multiple of 16 bytes for R = double and float)
//   FLOPS, scale()      kernel_functions.hpp FLOPS() / uKerScaleFactor
//   pack(rec, x, n, f)  build the LDS record of one source from the AoS inputs
//   pair<R,MODE,MASKED>(acc, d, rec, ctx, K)   one pair interaction, d = x_trg - x_src  (generic-kernel.txx:83)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "fastmath.hpp"

namespace sctl_amd {

constexpr double kPi = 3.141592653589793238462643383279502884;

// Kernel context passed by value in the launch arguments (replaces the host ctx_ptr, generic-kernel.hpp:150).
struct KerCtx { double v[4]; };

// ---- masked reciprocal square root ---------------------------------------------------------------------
// Semantics of approx_rsqrt<digits>(r2, r2 > 0) (vec.txx:361-364, intrin-wrapper.hpp:539-555): 0 where
// r2 == 0, else r2^-1/2 to `digits` digits.  The seed is the hardware v_rsq_f64 / v_rsq_f32 (measured max
// relative error 2^-24.2 / 2^-23.3, tools/ubench/rsq_accuracy.hip); v_rsq(0) = +inf, whose high word is
// replaced by 0 with two 32-bit VALU ops (the low word of +inf is already 0).  Refinement, with
// e = 1 - x y^2 computed by one FMA so that it is exact to fp64 rounding:
//   MODE 0: seed only                                (>= 7 digits)
//   MODE 1: one Newton step,  y + y e / 2            (error 3/8 e^2 <= 4.3e-15, rms 3.5e-16: >= 14 digits)
//   MODE 2: one Halley step,  y + y e (1/2 + 3/8 e)  (error O(e^3): rounding only)
// With y = 0 every refinement returns 0, so the mask survives; NaN/inf inputs propagate as in the reference.
// MASKED = false skips the two mask instructions (9.5 % of the Laplace kernel's time, tools/ubench/laplace_variants):
// a coincident pair then yields inf/NaN, which eval_kernel.hpp detects per LDS tile and repairs by re-running that
// tile with MASKED = true — the result is identical to the always-masked evaluation.
// The one constant that is not a hardware inline constant (3/8) lives in a VGPR pair for the whole kernel
// (RsqConst, made opaque to the optimiser so that it is not re-materialised with v_mov per use).
// (The two constants of the four-instruction cubic step rsqrt_cubic83 below: c = 1 + m 2^-25 with m = round(2/3 2^25), and k = 4 (c - 1) - (c - 1)^2,
// both exact doubles.)  They are wave-uniform and live in scalar registers: an fp64 VALU instruction takes one scalar operand for free, and the
// tile-centred kernel has no vector registers to spare.
constexpr double kCubicC = 0x1.aaaaaa8000000p+0, kCubicK = 0x1.1c71c6e38e38ep+1;
// (and of its forms for r^-3 and r^-5, rsqrt3_cubic / rsqrt5_cubic: c = 7/5, k = 28/75 and c = 9/7, k = 36/245, rounded the same way)
constexpr double kCubic3C = 0x1.6666660000000p+0, kCubic3K = 0x1.7e4b170a3d700p-2, kCubic5C = 0x1.4924920000000p+0, kCubic5K = 0x1.2cee3c14e5e00p-3;
template <class R> struct RsqConst {
  R c38, c53, k209, c3, k3, c5, k5;
  __device__ __forceinline__ RsqConst() : c38(R(0.375)), c53(R(kCubicC)), k209(R(kCubicK)), c3(R(kCubic3C)), k3(R(kCubic3K)), c5(R(kCubic5C)), k5(R(kCubic5K)) {
    asm volatile("" : "+v"(c38), "+s"(c53), "+s"(k209), "+s"(c3), "+s"(k3), "+s"(c5), "+s"(k5));
  }
};

template <int MODE, bool MASKED> __device__ __forceinline__ double rsqrt_masked(double r2, const RsqConst<double>& K) {
  double y = __builtin_amdgcn_rsq(r2);
  if (MASKED) {
  // r2 == 0 -> y = +inf = {hi 0x7ff00000, lo 0} -> 0: one 32-bit compare + one v_cndmask on the high word.
  // The empty asm only stops the optimiser from widening this into a 64-bit compare; the compare and select
  // themselves are compiler-generated so that the gfx950 trans->VALU hazard after v_rsq_f64 is padded correctly
  // (an asm block reading the v_rsq result directly is NOT padded and read a stale register: observed as a NaN).
  int hi = __double2hiint(y);
  asm("" : "+v"(hi));
  hi = (hi == 0x7ff00000) ? 0 : hi;
  y = __hiloint2double(hi, __double2loint(y));
  }
  if (MODE >= 1) {
    const double a = r2 * y;
    const double e = __builtin_fma(-a, y, 1.0);
    const double ye = y * e;
    if (MODE == 1) y = __builtin_fma(ye, 0.5, y);
    else y = __builtin_fma(ye, __builtin_fma(e, K.c38, 0.5), y);
  }
  return y;
}
template <int MODE, bool MASKED> __device__ __forceinline__ float rsqrt_masked(float r2, const RsqConst<float>&) {
  float y = __builtin_amdgcn_rsqf(r2);
  if (MASKED) y = (__float_as_uint(y) == 0x7f800000u) ? 0.0f : y;   // r2 == 0 -> +inf -> 0
  if (MODE >= 1) {   // one Newton step in fp32 (only when more than 7 digits are asked of fp32)
    const float a = r2 * y;
    const float e = __builtin_fmaf(-a, y, 1.0f);
    y = __builtin_fmaf(y * e, 0.5f, y);
  }
  return y;
}

// Per-kernel constants, constructed once at kernel entry and passed to every pair() call.
// A kernel may ask for LDS_DOUBLES doubles of workgroup scratch, which its Consts constructor fills (all lanes of the
// workgroup call it; it may synchronise).
template <class R> struct DefaultConsts {
  static constexpr int LDS_DOUBLES = 0;
  static constexpr bool HAS_VARIANT = false;   // true: pair<R, MODE, MASKED, VARIANT> exists and variant(ctx) picks it per launch (0 .. NUM_VARIANTS-1, default 2)
  RsqConst<R> rsq;
  __device__ __forceinline__ explicit DefaultConsts(double*) {}
  // hooks of the speculative (unmasked) tile pass of eval_kernel: a kernel whose fast path has a precondition records
  // violations per lane and reports them at the end of the tile, which is then re-run on the careful path
  __device__ __forceinline__ void begin_tile() const {}
  __device__ __forceinline__ bool tile_bad(const KerCtx&) const { return false; }
};
// The Helmholtz kernel runs on the modes' unnormalised reciprocal square roots too (rsqrt_scaled below: C / r with C = 2 for MODE 1, A = 2.6666666 for
// MODE 2): the amplitude's C goes into the scale, and the distance comes out as r2 (C / r) = C r at the price of the normalised one.  C by mode:
constexpr double helmholtz_dist_factor(int mode) { return mode == 1 ? 2.0 : mode == 2 ? 0x1.5555550000000p+1 : 1.0; }
template <class R> struct HelmholtzConsts;
template <> struct HelmholtzConsts<float> {          // fp32: libm sincosf / expf
  static constexpr int LDS_DOUBLES = 0;
  static constexpr bool HAS_VARIANT = true;
  static constexpr int NUM_VARIANTS = 2;
  __device__ __forceinline__ int variant(const KerCtx& ctx) const { return ctx.v[1] == 0 ? 1 : 0; }
  RsqConst<float> rsq;
  float cinv = 1.0f;       // 1 / C: pair() forms C r from the mode's unnormalised C / r (helmholtz_dist_factor)
  __device__ __forceinline__ explicit HelmholtzConsts(double*) {}
  __device__ __forceinline__ HelmholtzConsts(double*, int, const KerCtx&, int mode) : cinv((float)(1.0 / helmholtz_dist_factor(mode))) {}
  __device__ __forceinline__ void begin_tile() const {}
  __device__ __forceinline__ bool tile_bad(const KerCtx&) const { return false; }
};
template <> struct HelmholtzConsts<double> {         // fp64: table-driven e^{ikr} (fastmath.hpp), tables in LDS
  // Two table sets.  The all-pairs evaluator (256-lane workgroups that live for thousands of tiles) offers LDS_DOUBLES_ALL_PAIRS and, when
  // the wavenumber allows it (Re k > 0, |Im k| <= Re k / 4), runs the ONE-reduction form: 2048 complex nodes with the decay folded in +
  // 256 period factors (34 KB).  The one-wave evaluators (lists, matrix blocks) keep the small tables (10 KB) of the two-reduction form:
  // 34 KB per 64 lanes would cost them their occupancy.
  static constexpr int LDS_DOUBLES = fastmath::kTableDoubles;
  static constexpr int LDS_DOUBLES_ALL_PAIRS = fastmath::kCexpTableDoubles > fastmath::kTableDoubles ? fastmath::kCexpTableDoubles : fastmath::kTableDoubles;
  static constexpr bool HAS_VARIANT = true;
  static constexpr int NUM_VARIANTS = 4;               // 0: complex k, two reductions; 1: real k, sincos only; 2: complex k, one reduction; 3: real k, one reduction
  bool one_reduction;
  __device__ __forceinline__ int variant(const KerCtx& ctx) const { return (ctx.v[1] == 0 ? 1 : 0) + (one_reduction ? 2 : 0); }
  RsqConst<double> rsq;
  fastmath::TabCoeffsK tk;     // reduction and polynomial constants with the launch's wavenumber folded in: functions of the distance
  fastmath::CexpCoeffsK ck;    // the same for the one-reduction form
  const double* table;
  double cdist, cinv;          // C and 1 / C of the distance pair() hands over (C r)
  // (a Consts type constructible from (double*, int, const KerCtx&[, int mode]) is handed the scratch capacity, the launch's context and its accuracy mode: make_consts below)
  __device__ __forceinline__ HelmholtzConsts(double* lds, int lds_doubles, const KerCtx& ctx) : HelmholtzConsts(lds, lds_doubles, ctx, 0) {}
  __device__ __forceinline__ HelmholtzConsts(double* lds, int lds_doubles, const KerCtx& ctx, int mode) : table(lds), cdist(helmholtz_dist_factor(mode)), cinv(1.0 / helmholtz_dist_factor(mode)) {
    one_reduction = lds_doubles >= fastmath::kCexpTableDoubles && fastmath::CexpCoeffsK::usable(ctx.v[0], -ctx.v[1]);   // (+17 % over two reductions: profiles/r03_ab_helmholtz_one_reduction.txt)
    {
      const fastmath::Coeffs full;                   // the table-free polynomials, used here only
      if (one_reduction) fastmath::fill_cexp_tables(lds, (int)threadIdx.x, (int)blockDim.x, ctx.v[0], -ctx.v[1], full, fastmath::TabCoeffs());
      else fastmath::fill_tables(lds, (int)threadIdx.x, (int)blockDim.x, full);
    }
    __syncthreads();
    if (one_reduction) {
      ck.set_scaled(ctx.v[0], -ctx.v[1], fastmath::TabCoeffs(), cdist);   // the one-reduction form reduces C r directly
      ck.pin();
    } else {
      tk.set_scaled(ctx.v[0], -ctx.v[1], fastmath::TabCoeffs(), cdist);
      tk.pin();
    }
  }
  __device__ __forceinline__ HelmholtzConsts(double* lds, const KerCtx& ctx) : HelmholtzConsts(lds, LDS_DOUBLES, ctx) {}
  // the table-driven forms need |Re k| r <= kSincosTabMaxArg and |Im k| r <= kExpTabMaxArg (one reduction: Re k r <= kCexpMaxPhase): the
  // speculative pass only tracks the largest distance
  // (a distance is >= +0 or NaN, so its HIGH WORD orders like the number: the maximum is one 32-bit integer instruction per pair, and a
  // NaN's high word is larger than any finite one, which sends its tile to the careful pass)
  mutable unsigned rmax_hi = 0;
  __device__ __forceinline__ void begin_tile() const { rmax_hi = 0; }
  __device__ __forceinline__ void note_distance(double r) const {
    const unsigned h = (unsigned)__double2hiint(r);
    rmax_hi = (h > rmax_hi) ? h : rmax_hi;
  }
  __device__ __forceinline__ bool tile_bad(const KerCtx& ctx) const {
    const double rmax = __hiloint2double((int)(rmax_hi + 1u), 0);   // the next high word up bounds every low word
    if (one_reduction) return !(ctx.v[0] * rmax <= fastmath::kCexpMaxPhase * cdist);   // (the distances recorded are C r)
    return !(__builtin_fabs(ctx.v[0]) * rmax <= fastmath::kSincosTabMaxArg * cdist && __builtin_fabs(ctx.v[1]) * rmax <= fastmath::kExpTabMaxArg * cdist);
  }
};

// A kernel whose pair() leaves some entries of acc to be derived from others — a symmetric output: the traction kernel fills the upper
// triangle only — supplies `template <class R> static void finish(R (&acc)[K1])`, or `template <class R, int MODE> static void finish_mode(...)`
// when what is left to do depends on the accuracy mode (the fused Laplace kernel: its potential and its gradient accumulate different powers
// of the unnormalised 1/r); the evaluators apply it once, when the sums leave the registers (finish_acc).
template <class Ker, class R, int MODE, class = void, class = void> struct FinishOf {
  static __device__ __forceinline__ void apply(R (&)[Ker::K1]) {}
};
template <class Ker, class R, int MODE, class V> struct FinishOf<Ker, R, MODE, std::void_t<decltype(&Ker::template finish<R>)>, V> {
  static __device__ __forceinline__ void apply(R (&acc)[Ker::K1]) { Ker::template finish<R>(acc); }
};
template <class Ker, class R, int MODE> struct FinishOf<Ker, R, MODE, void, std::void_t<decltype(&Ker::template finish_mode<R, MODE>)>> {
  static __device__ __forceinline__ void apply(R (&acc)[Ker::K1]) { Ker::template finish_mode<R, MODE>(acc); }
};
template <class Ker, class R, int MODE> __device__ __forceinline__ void finish_acc(R (&acc)[Ker::K1]) { FinishOf<Ker, R, MODE>::apply(acc); }

// The LDS record of one source: Ker::pack<R>, or `template <class R, int MODE> static void pack_mode(rec, x, n, f)` when the record depends on
// the accuracy mode (kernels that keep a density pre-multiplied by a power of the unnormalised 1/r's factor, below).
template <class Ker, class R, int MODE, class = void> struct PackOf {
  static __device__ __forceinline__ void apply(R* rec, const R* x, const R* n, const R* f) { Ker::template pack<R>(rec, x, n, f); }
};
template <class Ker, class R, int MODE> struct PackOf<Ker, R, MODE, std::void_t<decltype(&Ker::template pack_mode<R, MODE>)>> {
  static __device__ __forceinline__ void apply(R* rec, const R* x, const R* n, const R* f) { Ker::template pack_mode<R, MODE>(rec, x, n, f); }
};
template <class Ker, class R, int MODE> __device__ __forceinline__ void pack_record(R* rec, const R* x, const R* n, const R* f) { PackOf<Ker, R, MODE>::apply(rec, x, n, f); }

// Per-kernel constants of a launch: Consts(lds, capacity, ctx) when the type takes the scratch capacity (in doubles) and the context,
// Consts(lds, ctx) when it takes the context, Consts(lds) otherwise.
template <class KC> __device__ __forceinline__ KC make_consts(double* lds, int lds_doubles, const KerCtx& ctx) {
  if constexpr (std::is_constructible<KC, double*, int, const KerCtx&>::value) return KC(lds, lds_doubles, ctx);
  else if constexpr (std::is_constructible<KC, double*, const KerCtx&>::value) return KC(lds, ctx);
  else return KC(lds);
}
template <class KC> __device__ __forceinline__ KC make_consts(double* lds, const KerCtx& ctx) { return make_consts<KC>(lds, KC::LDS_DOUBLES, ctx); }
// ... and the accuracy mode of the launch, for a Consts type constructible from (double*, int, const KerCtx&, int mode): constants that depend on
// which reciprocal square root pair() will use (HelmholtzConsts: the table reduction takes the distance as C r)
template <class KC> __device__ __forceinline__ KC make_consts(double* lds, int lds_doubles, const KerCtx& ctx, int mode) {
  if constexpr (std::is_constructible<KC, double*, int, const KerCtx&, int>::value) return KC(lds, lds_doubles, ctx, mode);
  else return make_consts<KC>(lds, lds_doubles, ctx);
}
template <class KC> __device__ __forceinline__ KC make_consts(double* lds, const KerCtx& ctx, int mode) { return make_consts<KC>(lds, KC::LDS_DOUBLES, ctx, mode); }
// Scratch the all-pairs evaluator offers a kernel: LDS_DOUBLES_ALL_PAIRS when the Consts type names one, LDS_DOUBLES otherwise.
template <class KC, class = void> struct AllPairsScratch { static constexpr int value = KC::LDS_DOUBLES; };
template <class KC> struct AllPairsScratch<KC, std::void_t<decltype(KC::LDS_DOUBLES_ALL_PAIRS)>> { static constexpr int value = KC::LDS_DOUBLES_ALL_PAIRS; };
// Number of launch-uniform variants of pair(): NUM_VARIANTS when named, else 2 for a HAS_VARIANT kernel (pair<R, MODE, MASKED, bool>).
template <class KC, class = void> struct NumVariants { static constexpr int value = KC::HAS_VARIANT ? 2 : 1; };
template <class KC> struct NumVariants<KC, std::void_t<decltype(KC::NUM_VARIANTS)>> { static constexpr int value = KC::NUM_VARIANTS; };

template <class R> __device__ __forceinline__ R fma_(R a, R b, R c);
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <class R> __device__ __forceinline__ R len2(const R (&d)[3]) { return fma_(d[2], d[2], fma_(d[1], d[1], d[0] * d[0])); }
template <class R> __device__ __forceinline__ R dot3(const R (&d)[3], const R* v) { return fma_(d[2], v[2], fma_(d[1], v[1], d[0] * v[0])); }

// MODE 1 for kernels whose terms all carry the same power of 1/r: the Newton step without its halving, y0 (3 - r2 y0^2) = 2/r
// (>= 14 digits), three instructions instead of four.  The factor 2^p goes into the scale applied once per target
// (Ker::acc_factor): this is the accuracy ParticleFMM (10 digits) and BoundaryIntegralOp (tol 1e-10) ask for by default, so it is
// the form the reference's own callers run.  With the masked seed y0 = 0 the result is 0, so the r = 0 rule survives.
// The step always lands low: with d the seed's relative error it returns (2/r)(1 - 3/2 d^2 - ...), measured over 1.3e8 arguments
// (tools/ubench/rsq_refine_accuracy.hip) mean -1.725e-16, range [-4.27e-15, +2.5e-16].  A one-sided error is a bias, and a bias does not
// average out over a sum; its mean is folded into acc_factor (newton2_factor(p) for a kernel whose terms carry (2/r)^p), i.e. into the
// scale applied once per target: no instruction, the per-pair error keeps its width and loses its offset.  (Centring the RANGE instead —
// a factor 1 + 3/4 d_max^2 — would put the worst case at +-2.1e-15 but move the mean to +1.9e-15: rms 1.9e-15 against 3.0e-16.)
constexpr double kNewton2MeanErr = -1.725e-16;
constexpr double newton2_factor(int p) { return (p == 1 ? 2.0 : p == 3 ? 8.0 : 32.0) * (1.0 + p * kNewton2MeanErr); }
template <bool MASKED, class R> __device__ __forceinline__ R rsqrt_newton2(R r2, const RsqConst<R>& K) {
  const R y = rsqrt_masked<0, MASKED>(r2, K);
  const R a = r2 * y;
  return y * fma_(-a, y, R(3));
}

// MODE 2 (full precision, the default) for the same kernels: the CUBIC step without its normalisation, in FOUR instructions.
// With w = r2 y0^2 = 1 - e the Halley polynomial 1 + e/2 + 3/8 e^2 is 3/8 (w^2 - 10/3 w + 5) = 3/8 ((w - 5/3)^2 + 20/9): a monic
// quadratic in w, i.e. ONE FMA for z = r2 (y0 y0) - 5/3 and one for z z + 20/9 — the small quantity e is never formed —
//     y0 ((r2 y0^2 - 5/3)^2 + 20/9) = (8/3) / r (1 + 5/16 e^3 + ...),
// against five instructions for y0 + y0 e (1/2 + 3/8 e); the factor (8/3)^p goes into the scale applied once per target, as for MODE 1.
// The constants need not BE 5/3 and 20/9, only be consistent: with c - 1 = B/2 and k = A - (c - 1)^2 the polynomial is A + B e + e^2, and what
// matters is B / A = 1/2 to ~2^-31 (its error is multiplied by e <= 2^-23) and 1/A = 3/8 to ~2^-8 (multiplied by e^2).  So c - 1 is 2/3 rounded
// to 25 bits, A = 4 (c - 1) = 2.666666627 and k = A - (c - 1)^2 are then EXACT doubles, B / A is exactly 1/2, 1/A is 3/8 (1 + 1.5e-8), and
// the scale carries no rounding of the factor for p = 1 (cubic83_factor(3), (5) are A^3, A^5 correctly rounded).
// Accuracy (tools/ubench/rsq_refine_accuracy.hip, 1.3e8 arguments against long double, profiles/r03_rsq_refine_accuracy.txt): three roundings
// sit on the main path (y0 y0, the polynomial, the product) where Halley's correction term has one, so the worst case is that of the reference's
// own default approx_rsqrt (2.5 ulp) rather than Halley's 1.25 ulp, with a smaller rms than the reference's.
// With the masked seed y0 = 0 the result is 0; an unmasked coincident pair gives z = 0 * inf = NaN, which the speculative pass detects.
constexpr double cubic83_factor(int p) { return p == 1 ? 0x1.5555550000000p+1 : p == 3 ? 0x1.2f684af684bdep+4 : 0x1.0db20937d5dcdp+7; }
template <bool MASKED> __device__ __forceinline__ double rsqrt_cubic83(double r2, const RsqConst<double>& K) {
  const double y = rsqrt_masked<0, MASKED>(r2, K);
  const double z = __builtin_fma(r2, y * y, -K.c53);
  return y * __builtin_fma(z, z, K.k209);
}
// fp32 never runs MODE 2 (capi.hip: mode_for); kept consistent with the shared scale factor
template <bool MASKED> __device__ __forceinline__ float rsqrt_cubic83(float r2, const RsqConst<float>& K) { return rsqrt_masked<1, MASKED>(r2, K) * (float)cubic83_factor(1); }
// 1/r times the factor Ker::acc_factor(MODE) accounts for, for a kernel whose terms all carry the same power of 1/r
// (A/B against the five-instruction Halley step, one box: profiles/r03_ab_cubic83.txt, +11 % on the headline)
template <int MODE, bool MASKED, class R> __device__ __forceinline__ R rsqrt_scaled(R r2, const RsqConst<R>& K) {
  if constexpr (MODE == 1) return rsqrt_newton2<MASKED>(r2, K);
  else if constexpr (MODE == 2) return rsqrt_cubic83<MASKED>(r2, K);
  else return rsqrt_masked<0, MASKED>(r2, K);
}
constexpr double rsqrt_scaled_factor(int mode, int p) { return mode == 1 ? newton2_factor(p) : mode == 2 ? cubic83_factor(p) : 1; }
// r^-3 and r^-5 for the kernels that need only that power (double layer, gradient, stresslet, traction).  MODE 2 does not cube the refined 1/r:
// (1 - e)^(-3/2) = 1 + 3/2 e + 15/8 e^2 + ... and (1 - e)^(-5/2) = 1 + 5/2 e + 35/8 e^2 + ... are again monic quadratics in w = r2 y0^2 up to a
// factor, 15/8 ((w - 7/5)^2 + 28/75) and 35/8 ((w - 9/7)^2 + 36/245), so
//     y0 s ((r2 s - 7/5)^2 + 28/75) = (8/15) / r^3,      y0 s^2 ((r2 s - 9/7)^2 + 36/245) = (8/35) / r^5,      s = y0^2,
// in five and six instructions where the cubic step and its powers take six and seven, and with fewer roundings on the way.  Constants as for
// the first power: c - 1 = 2/5 (2/7) rounded to a 25-bit multiple of 3 (5), A = 4/3 (c - 1) (4/5 (c - 1)) and k = A - (c - 1)^2 exact doubles, so
// that the ratio of the polynomial's first two coefficients is exactly 3/2 (5/2) and the factor A needs no rounding.
constexpr double kCubic3A = 0x1.1111100000000p-1, kCubic5A = 0x1.d41d400000000p-3;
template <bool MASKED> __device__ __forceinline__ double rsqrt3_cubic(double r2, const RsqConst<double>& K) {
  const double y = rsqrt_masked<0, MASKED>(r2, K), s = y * y;
  const double z = __builtin_fma(r2, s, -K.c3);
  return (y * s) * __builtin_fma(z, z, K.k3);
}
template <bool MASKED> __device__ __forceinline__ double rsqrt5_cubic(double r2, const RsqConst<double>& K) {
  const double y = rsqrt_masked<0, MASKED>(r2, K), s = y * y;
  const double z = __builtin_fma(r2, s, -K.c5);
  return (y * (s * s)) * __builtin_fma(z, z, K.k5);
}
constexpr double rsqrt_pow_factor(int mode, int p) { return mode == 2 ? (p == 1 ? cubic83_factor(1) : p == 3 ? kCubic3A : kCubic5A) : rsqrt_scaled_factor(mode, p); }
// r^-P times rsqrt_pow_factor(MODE, P), P = 3 or 5
template <int MODE, int P, bool MASKED, class R> __device__ __forceinline__ R rsqrt_pow_scaled(R r2, const RsqConst<R>& K) {
  static_assert(P == 3 || P == 5, "powers 3 and 5");
  if constexpr (MODE == 2 && std::is_same<R, double>::value) return P == 3 ? rsqrt3_cubic<MASKED>(r2, K) : rsqrt5_cubic<MASKED>(r2, K);
  else if constexpr (MODE == 2) {   // fp32 never runs MODE 2 (capi.hip: mode_for); kept consistent with the shared scale factor
    const R y = rsqrt_masked<1, MASKED>(r2, K), y2 = y * y;
    return (P == 3 ? y2 * y : y2 * y2 * y) * R(rsqrt_pow_factor(2, P));
  } else {
    const R y = rsqrt_scaled<MODE, MASKED>(r2, K), y2 = y * y;
    return P == 3 ? y2 * y : y2 * y2 * y;
  }
}

// Kernels whose terms carry SEVERAL powers of 1/r — Stokeslet-like u = (f + (r.f) r / r^2) / r — can use the unnormalised y = C / r as well when
// the record keeps the density twice: f for the dot product, whose term then carries y^3 = C^3 / r^3, and C^2 f for the term in y alone.  One
// fp64 instruction per pair less, paid with K0 more reals per source in LDS; the factor C^3 goes into the scale.  rsqrt_scaled_c2 is C^2: 4
// (exact) for the Newton step, A^2 (A has 24 bits: exact) for the cubic step, 1 for the bare seed.
constexpr double rsqrt_scaled_c2(int mode) { return rsqrt_scaled_factor(mode, 1) == 1 ? 1.0 : mode == 1 ? 4.0 : cubic83_factor(1) * cubic83_factor(1); }

// ---- Laplace single layer: u = f / r          (kernel_functions.hpp:15-31) -------------------------------
struct Laplace3D_FxU {
  static constexpr int ID = 0, K0 = 1, K1 = 1, ND = 0, NREC = 4, FLOPS = 6;
  static constexpr const char* NAME = "Laplace3D-FxU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 1 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 1); }   // MODE 1 accumulates f (2/r), MODE 2 f (8/3)/r
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R*, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0];
  }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R rinv = rsqrt_scaled<MODE, MASKED>(len2(d), K.rsq);   // MODE 1: 2/r, MODE 2: (8/3)/r
    acc[0] = fma_(rec[3], rinv, acc[0]);
  }
};

// ---- Laplace double layer: u = (r.n) f / r^3   (kernel_functions.hpp:33-51); record holds n*f ------------
struct Laplace3D_DxU {
  static constexpr int ID = 1, K0 = 1, K1 = 1, ND = 3, NREC = 6, FLOPS = 14;
  static constexpr const char* NAME = "Laplace3D-DxU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 1 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_pow_factor(mode, 3); }   // MODE 1 accumulates (r.n f) (2/r)^3, MODE 2 (r.n f) (8/15) / r^3
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = n[0] * f[0]; rec[4] = n[1] * f[0]; rec[5] = n[2] * f[0];
  }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    acc[0] = fma_(dot3(d, rec + 3), rsqrt_pow_scaled<MODE, 3, MASKED>(len2(d), K.rsq), acc[0]);
  }
};

// ---- gradient of the Laplace single layer: u_j = f r_j / r^3, scale -1/(4 pi)   (kernel_functions.hpp:53-72)
struct Laplace3D_FxdU {
  static constexpr int ID = 2, K0 = 1, K1 = 3, ND = 0, NREC = 4, FLOPS = 11;
  static constexpr const char* NAME = "Laplace3D-FxdU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return -1 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_pow_factor(mode, 3); }   // MODE 1 accumulates f r (2/r)^3, MODE 2 f r (8/15) / r^3
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R*, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0];
  }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R t = rsqrt_pow_scaled<MODE, 3, MASKED>(len2(d), K.rsq) * rec[3];
    for (int j = 0; j < 3; j++) acc[j] = fma_(t, d[j], acc[j]);
  }
};

// ---- Stokeslet: u_j = f_j / r + (r.f) r_j / r^3, scale 1/(8 pi)   (kernel_functions.hpp:74-95) -----------
struct Stokes3D_FxU {
  static constexpr int ID = 3, K0 = 3, K1 = 3, ND = 0, NREC = 10, FLOPS = 23;
  static constexpr const char* NAME = "Stokes3D-FxU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 1 / (8 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 3); }   // pair() accumulates C^3 x the kernel value, C / r the mode's 1/r
  // record: x, f (for the dot product), C^2 f (the term in 1/r alone; rsqrt_scaled_c2)
  template <class R, int MODE> static __device__ __forceinline__ void pack_mode(R* rec, const R* x, const R*, const R* f) {
    const R c2 = R(rsqrt_scaled_c2(MODE));
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0]; rec[4] = f[1]; rec[5] = f[2]; rec[6] = c2 * f[0]; rec[7] = c2 * f[1]; rec[8] = c2 * f[2]; rec[9] = 0;
  }
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) { pack_mode<R, 0>(rec, x, n, f); }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R y = rsqrt_scaled<MODE, MASKED>(len2(d), K.rsq);                                  // C / r
    const R t = dot3(d, rec + 3) * (y * y);                                                  // C^2 (r.f) / r^2
    for (int j = 0; j < 3; j++) acc[j] = fma_(y, fma_(t, d[j], rec[6 + j]), acc[j]);         // C^3 (f_j + r_j (r.f) / r^2) / r: 1/r^3 is never formed
  }
};

// ---- stresslet: u_j = r_j (r.f)(r.n) / r^5, scale 3/(4 pi)   (kernel_functions.hpp:97-120) ----------------
struct Stokes3D_DxU {
  static constexpr int ID = 4, K0 = 3, K1 = 3, ND = 3, NREC = 10, FLOPS = 26;
  static constexpr const char* NAME = "Stokes3D-DxU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 3 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_pow_factor(mode, 5); }   // MODE 1 accumulates (...) (2/r)^5, MODE 2 (...) (8/35) / r^5
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) {
    for (int k = 0; k < 3; k++) { rec[k] = x[k]; rec[3 + k] = n[k]; rec[6 + k] = f[k]; }
    rec[9] = 0;
  }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R t = dot3(d, rec + 3) * dot3(d, rec + 6) * rsqrt_pow_scaled<MODE, 5, MASKED>(len2(d), K.rsq);
    for (int j = 0; j < 3; j++) acc[j] = fma_(t, d[j], acc[j]);
  }
};

// ---- traction tensor: u_{jk} = (r.f) r_j r_k / r^5, scale -3/(4 pi)   (kernel_functions.hpp:122-146) -------
struct Stokes3D_FxT {
  static constexpr int ID = 5, K0 = 3, K1 = 9, ND = 0, NREC = 6, FLOPS = 39;
  static constexpr const char* NAME = "Stokes3D-FxT";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return -3 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_pow_factor(mode, 5); }   // MODE 1 accumulates (...) (2/r)^5, MODE 2 (...) (8/35) / r^5
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R*, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0]; rec[4] = f[1]; rec[5] = f[2];
  }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R t = dot3(d, rec + 3) * rsqrt_pow_scaled<MODE, 5, MASKED>(len2(d), K.rsq);
    for (int j = 0; j < 3; j++) {       // u_jk = u_kj: the upper triangle only (6 FMAs instead of 9); finish() fills in the rest
      const R tj = t * d[j];
      for (int k = j; k < 3; k++) acc[j * 3 + k] = fma_(tj, d[k], acc[j * 3 + k]);
    }
  }
  template <class R> static __device__ __forceinline__ void finish(R (&acc)[K1]) { acc[3] = acc[1]; acc[6] = acc[2]; acc[7] = acc[5]; }
};

// ---- Stokeslet + source/sink: u_j = f_j / r + ((r.f) + f_3) r_j / r^3   (kernel_functions.hpp:148-172) ------
struct Stokes3D_FSxU {
  static constexpr int ID = 6, K0 = 4, K1 = 3, ND = 0, NREC = 10, FLOPS = 26;
  static constexpr const char* NAME = "Stokes3D-FSxU";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 1 / (8 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 3); }   // as the Stokeslet: C^3 x the kernel value
  template <class R, int MODE> static __device__ __forceinline__ void pack_mode(R* rec, const R* x, const R*, const R* f) {
    const R c2 = R(rsqrt_scaled_c2(MODE));
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0]; rec[4] = f[1]; rec[5] = f[2]; rec[6] = f[3]; rec[7] = c2 * f[0]; rec[8] = c2 * f[1]; rec[9] = c2 * f[2];
  }
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) { pack_mode<R, 0>(rec, x, n, f); }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R y = rsqrt_scaled<MODE, MASKED>(len2(d), K.rsq);                                                 // C / r
    const R t = fma_(d[2], rec[5], fma_(d[1], rec[4], fma_(d[0], rec[3], rec[6]))) * (y * y);               // C^2 ((r.f) + f_3) / r^2
    for (int j = 0; j < 3; j++) acc[j] = fma_(y, fma_(t, d[j], rec[7 + j]), acc[j]);
  }
};

// ---- velocity + pressure: Stokeslet and p = (r.f) / r^3   (kernel_functions.hpp:174-198) -------------------
struct Stokes3D_FxUP {
  static constexpr int ID = 7, K0 = 3, K1 = 4, ND = 0, NREC = 10, FLOPS = 26;
  static constexpr const char* NAME = "Stokes3D-FxUP";
  template <class R> using Consts = DefaultConsts<R>;
  static constexpr double scale() { return 1 / (8 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 3); }   // velocity as the Stokeslet; the pressure (r.f) / r^3 carries C^3 by itself
  template <class R, int MODE> static __device__ __forceinline__ void pack_mode(R* rec, const R* x, const R*, const R* f) {
    const R c2 = R(rsqrt_scaled_c2(MODE));
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0]; rec[4] = f[1]; rec[5] = f[2]; rec[6] = c2 * f[0]; rec[7] = c2 * f[1]; rec[8] = c2 * f[2]; rec[9] = 0;
  }
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) { pack_mode<R, 0>(rec, x, n, f); }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R y = rsqrt_scaled<MODE, MASKED>(len2(d), K.rsq);
    const R t = dot3(d, rec + 3) * (y * y);
    for (int j = 0; j < 3; j++) acc[j] = fma_(y, fma_(t, d[j], rec[6 + j]), acc[j]);
    acc[3] = fma_(t, y, acc[3]);
  }
};

// ---- NEW: Laplace single + double layer -> potential and gradient (BASELINE config 2) -----
//   u      = q / r + mu (r.n) / r^3
//   grad u = -q r / r^3 + mu ( n / r^3 - 3 (r.n) r / r^5 ),    scale 1/(4 pi);  record holds m = mu * n and q.
template <class R> struct FDxUdUConsts : DefaultConsts<R> {   // + the constant 3 in a register for the whole kernel (no inline constant; re-made per source otherwise)
  R c3;
  __device__ __forceinline__ explicit FDxUdUConsts(double* lds) : DefaultConsts<R>(lds), c3(R(3)) { asm volatile("" : "+v"(c3)); }
};
struct Laplace3D_FDxUdU {
  static constexpr int ID = 8, K0 = 2, K1 = 4, ND = 3, NREC = 10, FLOPS = 28;
  static constexpr const char* NAME = "Laplace3D-FDxUdU";
  template <class R> using Consts = FDxUdUConsts<R>;
  static constexpr double scale() { return 1 / (4 * kPi); }
  // With y = C / r (the mode's unnormalised 1/r) the gradient (m_j - r_j (q + 3 mu (r.n) / r^2)) / r^3 accumulates C^5 x its value when m and q are
  // kept as C^2 m, C^2 q beside the plain m of the dot product; the potential (q + mu (r.n) / r^2) / r then carries C^3 and is multiplied by C^2
  // once per target, when the sums leave the registers (finish_mode).
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 5); }
  // record: x, m = mu n (for the dot product), C^2 m, C^2 q
  template <class R, int MODE> static __device__ __forceinline__ void pack_mode(R* rec, const R* x, const R* n, const R* f) {
    const R c2 = R(rsqrt_scaled_c2(MODE));
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2];
    rec[3] = n[0] * f[1]; rec[4] = n[1] * f[1]; rec[5] = n[2] * f[1];
    rec[6] = c2 * rec[3]; rec[7] = c2 * rec[4]; rec[8] = c2 * rec[5]; rec[9] = c2 * f[0];
  }
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R* n, const R* f) { pack_mode<R, 0>(rec, x, n, f); }
  template <class R, int MODE, bool MASKED> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx&, const Consts<R>& K) {
    const R y = rsqrt_scaled<MODE, MASKED>(len2(d), K.rsq);   // C / r
    const R y2 = y * y;
    const R y3 = y2 * y;
    const R w = dot3(d, rec + 3) * y2;                      // C^2 mu (r.n) / r^2
    acc[0] = fma_(w, y, fma_(rec[9], y, acc[0]));           // C^3 (q / r + mu (r.n) / r^3)
    const R c = fma_(w, K.c3, rec[9]);                      // C^2 (q + 3 mu (r.n) / r^2)
    for (int j = 0; j < 3; j++) acc[1 + j] = fma_(y3, fma_(-d[j], c, rec[6 + j]), acc[1 + j]);   // C^5 (m_j - r_j c) / r^3
  }
  template <class R, int MODE> static __device__ __forceinline__ void finish_mode(R (&acc)[K1]) { acc[0] *= R(rsqrt_scaled_c2(MODE)); }
};

static_assert(helmholtz_dist_factor(2) == rsqrt_scaled_factor(2, 1) && helmholtz_dist_factor(0) == 1, "HelmholtzConsts' distance factor is the cubic step's A");
// ---- NEW: Helmholtz single layer G = exp(ikr)/r, complex k = ctx.v[0] + i ctx.v[1] (config 5)
//   (u_re, u_im) += G (f_re, f_im) as complex numbers, scale 1/(4 pi), G = 0 at r = 0.
struct Helmholtz3D_FxU {
  static constexpr int ID = 9, K0 = 2, K1 = 2, ND = 0, NREC = 6, FLOPS = 16;
  static constexpr const char* NAME = "Helmholtz3D-FxU";
  template <class R> using Consts = HelmholtzConsts<R>;
  static constexpr double scale() { return 1 / (4 * kPi); }
  static constexpr double acc_factor(int mode) { return rsqrt_scaled_factor(mode, 1); }   // the amplitude is the mode's C / r
  template <class R> static __device__ __forceinline__ void pack(R* rec, const R* x, const R*, const R* f) {
    rec[0] = x[0]; rec[1] = x[1]; rec[2] = x[2]; rec[3] = f[0]; rec[4] = f[1]; rec[5] = 0;
  }
  // VARIANT (launch-uniform, Consts::variant): bit 0 = real wavenumber (no decay factor), bit 1 = the one-reduction form (fp64 all-pairs only)
  template <class R, int MODE, bool MASKED, int VARIANT = 0> static __device__ __forceinline__ void pair(R (&acc)[K1], const R (&d)[3], const R* rec, const KerCtx& ctx, const Consts<R>& K) {
    constexpr bool REAL_K = (VARIANT & 1) != 0;
    const R r2 = len2(d);
    const R rinv = rsqrt_scaled<MODE, MASKED>(r2, K.rsq);   // C / r: the amplitude, its C in the scale
    const R rs = r2 * rinv;                                  // C r
    if constexpr ((VARIANT & 2) != 0) {
      R gr, gi;
      cexp_<MASKED, REAL_K>(rs, rinv, ctx, gr, gi, K);        // reduces C r with the constants of k / C (HelmholtzConsts)
      acc[0] = fma_(gr, rec[3], fma_(-gi, rec[4], acc[0]));
      acc[1] = fma_(gi, rec[3], fma_(gr, rec[4], acc[1]));
      return;
    }
    const R r = std::is_same<R, double>::value ? rs : rs * R(K.cinv);   // fp64: the table forms reduce C r with the constants of k / C; fp32 (hardware sine / cosine / exp2) takes r
    R sn, cs;
    sincos_<MASKED>(r, ctx, sn, cs, K);
    R amp = rinv;
    if (!REAL_K) amp *= exp_<MASKED>(r, ctx, K);
    const R gr = amp * cs, gi = amp * sn;
    acc[0] = fma_(gr, rec[3], fma_(-gi, rec[4], acc[0]));
    acc[1] = fma_(gi, rec[3], fma_(gr, rec[4], acc[1]));
  }
  // fp64, one reduction of r for the whole factor e^{ikr} (fastmath.hpp: cexp_tab_k); returns G = e^{ikr} / r.  The speculative pass runs it
  // unconditionally and records the largest distance; the careful pass branches per pair to libm beyond the table's range.
  // (r is C r and rinv C / r here, C = K.cdist)
  template <bool MASKED, bool REAL_K> static __device__ __forceinline__ void cexp_(double r, double rinv, const KerCtx& ctx, double& gr, double& gi, const HelmholtzConsts<double>& K) {
    if (MASKED && __builtin_expect(!(ctx.v[0] * r <= fastmath::kCexpMaxPhase * K.cdist), 0)) {
      double s, c;
      const double rt = r * K.cinv;
      ::sincos(ctx.v[0] * rt, &s, &c);
      const double amp = REAL_K ? rinv : rinv * ::exp(-ctx.v[1] * rt);
      gr = amp * c; gi = amp * s;
      return;
    }
    if (!MASKED) K.note_distance(r);
    double re, im;
    if (REAL_K) {
      fastmath::cexp_tab_k_real(r, re, im, K.ck, K.table);
      gr = rinv * re; gi = rinv * im;
    } else {
      double dm;
      fastmath::cexp_tab_k(r, re, im, dm, K.ck, K.table);
      const double amp = rinv * dm;
      gr = amp * re; gi = amp * im;
    }
  }
  template <bool MASKED, bool REAL_K> static __device__ __forceinline__ void cexp_(float, float, const KerCtx&, float&, float&, const HelmholtzConsts<float>&) {}   // fp32 has no such variant
  // fp64: Cody-Waite reduction to the nearest node of an LDS table + a short polynomial (fastmath.hpp).  The speculative pass
  // (MASKED = false) runs the forms with the wavenumber folded in, unconditionally, as straight-line code, and only records the
  // largest distance (one v_max_f64) for the end-of-tile check; a tile with |Re k| r > 1.2e4 (two thousand wavelengths) or
  // |Im k| r > 700 is re-run by the careful pass, which branches per pair (libm beyond the table's range).
  template <bool MASKED> static __device__ __forceinline__ void sincos_(double r, const KerCtx& ctx, double& s, double& c, const HelmholtzConsts<double>& K) {
    if (!MASKED) {
      K.note_distance(r);
      fastmath::sincos_tab_k(r, s, c, K.tk, K.table);
      return;
    }
    const double x = ctx.v[0] * r;                       // (r is C r here: HelmholtzConsts)
    if (__builtin_expect(__builtin_fabs(x) > fastmath::kSincosTabMaxArg * K.cdist, 0)) ::sincos(x * K.cinv, &s, &c);
    else fastmath::sincos_tab_k(r, s, c, K.tk, K.table);
  }
  // fp32: the hardware's sine / cosine / exp2 (v_sin_f32, v_cos_f32, v_exp_f32: ~1e-6 absolute, inputs in revolutions / powers of
  // two) behind a two-piece reduction, so that the reduced argument keeps 24 bits however many periods x spans.  The phase
  // error that remains is the rounding of x itself (|x| 6e-8), which libm's exact reduction cannot remove either.
  template <bool MASKED> static __device__ __forceinline__ void sincos_(float r, const KerCtx& ctx, float& s, float& c, const HelmholtzConsts<float>&) {
    const float x = float(ctx.v[0]) * r;
    const float c_hi = 0.15915494f, c_lo = 6.4206383e-9f;       // 1/(2 pi) = c_hi + c_lo
    const float n = __builtin_rintf(x * c_hi);
    float fr = __builtin_fmaf(x, c_hi, -n);
    fr = __builtin_fmaf(x, c_lo, fr);
    s = __builtin_amdgcn_sinf(fr);
    c = __builtin_amdgcn_cosf(fr);
  }
  // exp(-Im k r): r >= 0 is clamped to 800/|Im k| (one v_min_f64; a NaN distance turns into the cap, but then rinv is NaN
  // too and the product stays NaN), so the argument handed to the table code is within its +-800 range
  template <bool MASKED> static __device__ __forceinline__ double exp_(double r, const KerCtx& ctx, const HelmholtzConsts<double>& K) {
    if (!MASKED) return fastmath::exp_tab_k(r, K.tk, K.table);   // range checked at the end of the tile (tile_bad)
    // careful pass: r >= 0 clamped to kExpTabMaxArg / |Im k| — the value there is 0 or inf already (a NaN distance turns into the cap, but
    // then rinv is NaN too and the product stays NaN)
    return fastmath::exp_tab_k(__builtin_fmin(r, fastmath::kExpTabMaxArg * K.cdist / __builtin_fabs(ctx.v[1])), K.tk, K.table);
  }
  template <bool MASKED> static __device__ __forceinline__ float exp_(float r, const KerCtx& ctx, const HelmholtzConsts<float>&) {
    const float x = -float(ctx.v[1]) * r;
    const float l_hi = 1.4426950f, l_lo = 1.9259630e-8f;        // log2(e) = l_hi + l_lo
    const float n = __builtin_fminf(__builtin_fmaxf(__builtin_rintf(x * l_hi), -300.0f), 300.0f);
    float fr = __builtin_fmaf(x, l_hi, -n);
    fr = __builtin_fmaf(x, l_lo, fr);
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(fr), (int)n);   // |x| beyond the clamp: fr is huge, exp2 gives 0 / inf as it should
  }
};

}  // namespace sctl_amd
