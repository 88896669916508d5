// fp64 sincos and exp for the Helmholtz kernel, written as straight-line fp64 FMA code (no tables, no divergent
// branches in the common range) so that a wave64 spends ~26 + ~18 fp64 issue slots instead of the ~80 of the generic
// device libm calls.  Host-and-device functions: tests/cpp/fastmath_check.cpp verifies them on the CPU against libm
// (max error ~1 ulp of the result's magnitude scale), the GPU parity tests cover the device build.
//
// The reference offers the CPU counterparts approx_sincos / approx_exp (include/sctl/vec.hpp:380-384,
// include/sctl/intrin-wrapper.hpp:640-778); nothing of their implementation is used here.
//   sincos: Cody-Waite reduction by pi/2 in three pieces (exact products for |n| < 2^20), degree-13/14 minimax
//           polynomials on [-pi/4, pi/4] (the classic fdlibm kernel coefficients), quadrant fix-up by sign/swap.
//           |x| > 1.6e6 takes the libm path (wave-uniform branch; never taken for k r of physical interest).
//   exp:    n = rint(x log2 e), r = x - n ln2 (two pieces), degree-13 Taylor/Horner on |r| <= ln2/2, scale by 2^n.
#pragma once
#include <cmath>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SCTL_AMD_HD __host__ __device__ __forceinline__
#else
#define SCTL_AMD_HD inline
#endif

namespace sctl_amd {
namespace fastmath {

SCTL_AMD_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// All non-inline constants of sincos/exp.  On the device the kernel constructs ONE Coeffs at entry and pins every
// member in a scalar register pair (pin()), so each polynomial step is a single v_fma_f64 with an SGPR operand; left to
// itself hipcc re-materialises the 25 literals with v_mov_b64 per use (24 extra VALU instructions per pair).
struct Coeffs {
  double S[6], C[6], two_over_pi, pio2[3], log2e, ln2hi, ln2lo, E[12];
  SCTL_AMD_HD Coeffs() {
    S[0] = -1.66666666666666324348e-01; S[1] = 8.33333333332248946124e-03; S[2] = -1.98412698298579493134e-04;
    S[3] = 2.75573137070700676789e-06; S[4] = -2.50507602534068634195e-08; S[5] = 1.58969099521155010221e-10;
    C[0] = 4.16666666666666019037e-02; C[1] = -1.38888888888741095749e-03; C[2] = 2.48015872894767294178e-05;
    C[3] = -2.75573143513906633035e-07; C[4] = 2.08757232129817482790e-09; C[5] = -1.13596475577881948265e-11;
    two_over_pi = 6.36619772367581382433e-01;
    pio2[0] = -1.57079632673412561417e+00;   // -(pi/2), first 33 bits
    pio2[1] = -6.07710050630396597660e-11;   // next 33 bits
    pio2[2] = -2.02226624879595063154e-21;   // tail
    log2e = 1.44269504088896338700e+00;
    ln2hi = -6.93147180369123816490e-01;
    ln2lo = -1.90821492927058770002e-10;
    E[0] = 1.6666666666666666e-01;  E[1] = 4.1666666666666664e-02;  E[2] = 8.3333333333333332e-03;  E[3] = 1.3888888888888889e-03;   // 1/3! ..
    E[4] = 1.9841269841269841e-04;  E[5] = 2.4801587301587302e-05;  E[6] = 2.7557319223985888e-06;  E[7] = 2.7557319223985893e-07;
    E[8] = 2.5052108385441720e-08;  E[9] = 2.0876756987868100e-09;  E[10] = 1.6059043836821613e-10; E[11] = 1.1470745597729725e-11; // .. 1/14!
  }
#ifdef __HIPCC__
  __device__ __forceinline__ void pin() {
    for (int i = 0; i < 6; i++) { asm volatile("" : "+s"(S[i])); asm volatile("" : "+s"(C[i])); }
    for (int i = 0; i < 3; i++) asm volatile("" : "+s"(pio2[i]));
    for (int i = 0; i < 12; i++) asm volatile("" : "+s"(E[i]));
    asm volatile("" : "+s"(two_over_pi)); asm volatile("" : "+s"(log2e)); asm volatile("" : "+s"(ln2hi)); asm volatile("" : "+s"(ln2lo));
  }
#endif
};

// sin(y), cos(y) for |y| <= pi/4 (+ a little): kernel polynomials
SCTL_AMD_HD void sincos_kernel(double y, double& s, double& c, const Coeffs& K) {
  const double z = y * y;
  double ps = K.S[5];
  ps = fma_(ps, z, K.S[4]);
  ps = fma_(ps, z, K.S[3]);
  ps = fma_(ps, z, K.S[2]);
  ps = fma_(ps, z, K.S[1]);
  ps = fma_(ps, z, K.S[0]);
  s = fma_(y * z, ps, y);
  double pc = K.C[5];
  pc = fma_(pc, z, K.C[4]);
  pc = fma_(pc, z, K.C[3]);
  pc = fma_(pc, z, K.C[2]);
  pc = fma_(pc, z, K.C[1]);
  pc = fma_(pc, z, K.C[0]);
  c = fma_(z * z, pc, fma_(z, -0.5, 1.0));
}

constexpr double kSincosMaxArg = 1.6e6;   // n < 2^20: the three-piece reduction below stays accurate

// precondition: |x| <= kSincosMaxArg (the caller routes larger arguments to libm)
SCTL_AMD_HD void sincos_reduced(double x, double& s, double& c, const Coeffs& K) {
  const double n = __builtin_rint(x * K.two_over_pi);
  double y = fma_(n, K.pio2[0], x);
  y = fma_(n, K.pio2[1], y);
  y = fma_(n, K.pio2[2], y);
  double sk, ck;
  sincos_kernel(y, sk, ck, K);
  const int q = (int)n;
  const bool swap = (q & 1) != 0;
  const double s0 = swap ? ck : sk, c0 = swap ? sk : ck;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

SCTL_AMD_HD double exp_fast(double x, const Coeffs& K) {
  // clamp so that n fits an int and ldexp saturates to 0 / inf correctly
  const double xc = __builtin_fmin(__builtin_fmax(x, -800.0), 800.0);   // v_max_f64 / v_min_f64; a NaN is restored below
  const double n = __builtin_rint(xc * K.log2e);
  double r = fma_(n, K.ln2hi, xc);
  r = fma_(n, K.ln2lo, r);
  double p = K.E[10];                       // 1/13!
#if defined(__clang__)
#pragma unroll
#endif
  for (int i = 9; i >= 0; i--) p = fma_(p, r, K.E[i]);
  p = fma_(p, r, 0.5);
  p = fma_(p * r, r, r) + 1.0;              // 1 + r + r^2 p
  const double e = __builtin_ldexp(p, (int)n);
  return (x != x) ? x : e;                  // NaN in -> NaN out
}

// ---- table-driven forms (what the device kernel's inner loop runs) ------------------------------------------------------
// A workgroup fills two small LDS tables once, with the polynomial code above, and every pair then needs only a short
// polynomial in the distance to the nearest table node:
//   sincos: nodes j pi/256, j = 0..511 (one full period: no quadrant logic), x = n pi/256 + y, |y| <= pi/512;
//           sin x = S_j + (S_j (cos y - 1) + C_j sin y), cos x = C_j + (C_j (cos y - 1) - S_j sin y), with
//           sin y = y + y z (s1 + s2 z), cos y - 1 = z (c1 + c2 z), z = y^2 (truncation < 8e-17).  Two-piece reduction
//           (33-bit head, exact product for |n| < 2^20, i.e. |x| < 1.2e4): the ABSOLUTE error stays ~1e-16, which is what a
//           kernel value cos + i sin of unit modulus needs; larger arguments take the libm path in the caller.
//   exp:    nodes 2^(j/256), x = n ln2/256 + r, |r| <= ln2/512, e^r - 1 by a degree-5 polynomial (degree 4 in the folded form below).
// 16 + 15 fp64 issue slots against 28 + 24 for the table-free code; 14 + 11 for the forms with the wavenumber folded in (below).
constexpr int kTrigNodes = 512, kExpShift = 8, kExpNodes = 1 << kExpShift;
constexpr int kTableDoubles = 2 * kTrigNodes + kExpNodes;   // [sin_j, cos_j] pairs, then 2^(j/256)
constexpr double kSincosTabMaxArg = 1.2e4;
constexpr double kExpTabMaxArg = 1.0e6;   // exp_tab_k: the integer part 256 x / ln2 must stay within 31 bits; from |x| = 746 on the result is 0 or inf

struct TabCoeffs {
  double inv_h, h1, h2, s1, s2, c1, c2, inv_e, e1, e2, p2, p3, p4;
  SCTL_AMD_HD TabCoeffs() {
    inv_h = 8.14873308630504119e+01;                      // 256/pi
    h1 = -1.57079632673412561417e+00 / 128;               // -(pi/256), first 33 bits (fdlibm's pio2_1, scaled exactly)
    h2 = -6.07710050650619224932e-11 / 128;               // -(pi/256 - head)
    s1 = -1.66666666666666666667e-01; s2 = 8.33333333333333333333e-03;
    c1 = -0.5; c2 = 4.16666666666666666667e-02;
    inv_e = kExpNodes * 1.44269504088896338700e+00;       // 256 log2(e)
    e1 = -6.93147180369123816490e-01 / kExpNodes;         // -(ln2/256), head with 32 significant bits
    e2 = -1.90821492927058770002e-10 / kExpNodes;
    p2 = 1.66666666666666666667e-01; p3 = 4.16666666666666666667e-02; p4 = 8.33333333333333333333e-03;   // 1/3!, 1/4!, 1/5!
  }
#ifdef __HIPCC__
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+s"(inv_h)); asm volatile("" : "+s"(h1)); asm volatile("" : "+s"(h2)); asm volatile("" : "+s"(s1)); asm volatile("" : "+s"(s2));
    asm volatile("" : "+s"(c2)); asm volatile("" : "+s"(inv_e)); asm volatile("" : "+s"(e1)); asm volatile("" : "+s"(e2));
    asm volatile("" : "+s"(p2)); asm volatile("" : "+s"(p3)); asm volatile("" : "+s"(p4));
  }
#endif
};

// Table entries, each from an argument reduced EXACTLY in integers (no rounding of j pi/256 beyond one multiplication
// of a number <= pi/4): node j = quadrant q, offset m/128 of a quadrant, folded to |angle| <= pi/4.
SCTL_AMD_HD void trig_node(int j, double& s, double& c, const Coeffs& K) {
  int q = (j >> 7) & 3, m = j & 127;
  if (m > 64) { m -= 128; q = (q + 1) & 3; }
  double sk, ck;
  sincos_kernel(m * (3.14159265358979323846 / 256), sk, ck, K);
  const double s0 = (q & 1) ? ck : sk, c0 = (q & 1) ? sk : ck;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
SCTL_AMD_HD double exp2_node(int j, const Coeffs& K) { return j == 0 ? 1.0 : exp_fast(j * (6.93147180559945309417e-01 / kExpNodes), K); }

// fill table[kTableDoubles] cooperatively: lane `tid` of `nthreads`
SCTL_AMD_HD void fill_tables(double* table, int tid, int nthreads, const Coeffs& K) {
  for (int j = tid; j < kTrigNodes; j += nthreads) trig_node(j, table[2 * j], table[2 * j + 1], K);
  for (int j = tid; j < kExpNodes; j += nthreads) table[2 * kTrigNodes + j] = exp2_node(j, K);
}

// Round-to-nearest-integer by the magic-number addition: t = a b + 1.5 2^52 has unit ulp, so t - magic is rint(a b) and the
// low dword of t holds that integer in two's complement (|a b| < 2^31) -- one fma + one add instead of mul + rint + cvt.
constexpr double kRoundMagic = 6755399441055744.0;
SCTL_AMD_HD int low_dword(double t) {
  long long b;
  __builtin_memcpy(&b, &t, 8);
  return (int)(unsigned)(unsigned long long)b;
}

// precondition: |x| <= kSincosTabMaxArg
SCTL_AMD_HD void sincos_tab(double x, double& s, double& c, const TabCoeffs& K, const double* table) {
  const double t = fma_(x, K.inv_h, kRoundMagic);
  const double n = t - kRoundMagic;
  double y = fma_(n, K.h1, x);
  y = fma_(n, K.h2, y);
  const int j = low_dword(t) & (kTrigNodes - 1);
  const double sj = table[2 * j], cj = table[2 * j + 1];
  const double z = y * y;
  const double sy = fma_(y * z, fma_(z, K.s2, K.s1), y);
  const double cm1 = z * fma_(z, K.c2, K.c1);
  s = fma_(sj, cm1, fma_(cj, sy, sj));
  c = fma_(cj, cm1, fma_(-sj, sy, cj));
}

// x must lie in [-800, 800] (the caller clamps: beyond that the result is 0 or inf anyway); NaN in -> NaN out
SCTL_AMD_HD double exp_tab_clamped(double xc, const TabCoeffs& K, const double* table) {
  const double tm = fma_(xc, K.inv_e, kRoundMagic);
  const double n = tm - kRoundMagic;
  double r = fma_(n, K.e1, xc);
  r = fma_(n, K.e2, r);
  const int ni = low_dword(tm);
  const double t = table[2 * kTrigNodes + (ni & (kExpNodes - 1))];
  double p = fma_(K.p4, r, K.p3);
  p = fma_(p, r, K.p2);
  p = fma_(p, r, 0.5);
  const double em1 = fma_(r * r, p, r);
  return __builtin_ldexp(fma_(t, em1, t), ni >> kExpShift);
}

// ---- the same two functions of x = k r, with the constant k folded into the reduction and the polynomial --------------------------
// The Helmholtz kernel needs sincos(kr r) and exp(kappa r) for ONE wavenumber per launch and a distance r per pair.  Forming
// x = k r first costs a multiplication per pair and function; instead the period is divided by k once per launch:
//   r = n (pi/256)/kr + y,  sin(kr y) = y (kr + z (S1 + S2 z)),  cos(kr y) - 1 = z (C1 + C2 z),  z = y^2,  S1 = -kr^3/6, ...
//   r = n (ln2/256)/kappa + q,  e^(kappa q) - 1 = q (P1 + q (P2 + q (P3 + q P4))),  P_m = kappa^m / m!   (|kappa q| <= ln2/512: truncation 3.8e-17)
// (the quotients as two-piece values, so that the reduced argument is as exact as before).  The rounding of x = k r itself
// disappears; everything else is operation for operation the code above.  k = 0 gives n = 0 and the value at 0.
struct TabCoeffsK {
  double ih, h1, h2, s0, s1, s2, c1, c2;    // sincos(kr r)
  double ie, e1, e2, p1, p2, p3, p4;        // exp(kappa r)
  // (hi + lo) / d as a two-piece quotient
  static SCTL_AMD_HD void div2(double hi, double lo, double d, double& q1, double& q2) {
    q1 = hi / d;
    q2 = (fma_(-q1, d, hi) + lo) / d;
  }
  SCTL_AMD_HD void set(double kr, double kappa, const TabCoeffs& B) {
    ih = B.inv_h * kr;
    if (kr != 0) div2(B.h1, B.h2, kr, h1, h2); else h1 = h2 = 0;
    const double k2 = kr * kr;
    s0 = kr; s1 = B.s1 * k2 * kr; s2 = B.s2 * k2 * k2 * kr;
    c1 = B.c1 * k2; c2 = B.c2 * k2 * k2;
    ie = B.inv_e * kappa;
    if (kappa != 0) div2(B.e1, B.e2, kappa, e1, e2); else e1 = e2 = 0;
    const double a2 = kappa * kappa;
    p1 = kappa; p2 = 0.5 * a2; p3 = B.p2 * a2 * kappa; p4 = B.p3 * a2 * a2;
  }
#ifdef __HIPCC__
  // the values were computed by vector instructions (the same in every lane): move them to scalar registers first
  static __device__ __forceinline__ void to_sgpr(double& v) {
    v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  }
  __device__ __forceinline__ void pin() {
    to_sgpr(ih); to_sgpr(h1); to_sgpr(h2); to_sgpr(s0); to_sgpr(s1); to_sgpr(s2); to_sgpr(c1); to_sgpr(c2);
    to_sgpr(ie); to_sgpr(e1); to_sgpr(e2); to_sgpr(p1); to_sgpr(p2); to_sgpr(p3); to_sgpr(p4);
  }
#endif
};

// sincos(kr r); precondition |kr| r <= kSincosTabMaxArg
SCTL_AMD_HD void sincos_tab_k(double r, double& s, double& c, const TabCoeffsK& K, const double* table) {
  const double t = fma_(r, K.ih, kRoundMagic);
  const double n = t - kRoundMagic;
  double y = fma_(n, K.h1, r);
  y = fma_(n, K.h2, y);
  const int j = low_dword(t) & (kTrigNodes - 1);
  const double sj = table[2 * j], cj = table[2 * j + 1];
  const double z = y * y;
  const double sy = y * fma_(z, fma_(z, K.s2, K.s1), K.s0);
  const double cm1 = z * fma_(z, K.c2, K.c1);
  s = fma_(sj, cm1, fma_(cj, sy, sj));
  c = fma_(cj, cm1, fma_(-sj, sy, cj));
}

// exp(kappa r); precondition |kappa| r <= kExpTabMaxArg (the caller's end-of-tile check); 0 / inf beyond the double range, NaN in -> NaN out
SCTL_AMD_HD double exp_tab_k(double r, const TabCoeffsK& K, const double* table) {
  const double tm = fma_(r, K.ie, kRoundMagic);
  const double n = tm - kRoundMagic;
  double q = fma_(n, K.e1, r);
  q = fma_(n, K.e2, q);
  const int ni = low_dword(tm);
  const double t = table[2 * kTrigNodes + (ni & (kExpNodes - 1))];
  double p = fma_(K.p4, q, K.p3);
  p = fma_(p, q, K.p2);
  p = fma_(p, q, K.p1);
  return __builtin_ldexp(fma_(t, p * q, t), ni >> kExpShift);
}

SCTL_AMD_HD double exp_tab(double x, const TabCoeffs& K, const double* table) {
  const double e = exp_tab_clamped(__builtin_fmin(__builtin_fmax(x, -800.0), 800.0), K, table);
  return (x != x) ? x : e;                  // fmin/fmax drop a NaN: restore it
}

}  // namespace fastmath
}  // namespace sctl_amd
