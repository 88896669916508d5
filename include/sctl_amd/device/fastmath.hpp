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
    q1 = hi / d;   // (an explicit FMA below: contraction cannot change it)
    q2 = (fma_(-q1, d, hi) + lo) / d;
  }
  SCTL_AMD_HD void set(double kr, double kappa, const TabCoeffs& B) { set_scaled(kr, kappa, B, 1.0); }
  // the same for a distance handed in as C r (see CexpCoeffsK::set_scaled): the constants of k / C, the two steps C h kept in two exact pieces
  static SCTL_AMD_HD void mul2(double C, double hi, double lo, double& ph, double& pl) {
    ph = C * hi;
    pl = fma_(C, hi, -ph) + C * lo;
  }
  SCTL_AMD_HD void set_scaled(double kr_true, double kappa_true, const TabCoeffs& B, double C) {
    const double kr = kr_true / C, kappa = kappa_true / C;
    double ph, pl;
    ih = B.inv_h * kr;
    mul2(C, B.h1, B.h2, ph, pl);
    if (kr != 0) div2(ph, pl, kr_true, h1, h2); else h1 = h2 = 0;
    const double k2 = kr * kr;
    s0 = kr; s1 = B.s1 * k2 * kr; s2 = B.s2 * k2 * k2 * kr;
    c1 = B.c1 * k2; c2 = B.c2 * k2 * k2;
    ie = B.inv_e * kappa;
    mul2(C, B.e1, B.e2, ph, pl);
    if (kappa != 0) div2(ph, pl, kappa_true, e1, e2); else e1 = e2 = 0;
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

// ---- ONE reduction of the distance for the whole factor e^{i k r}, k = kr + i ki (round 3) ---------------------------------------
// sincos_tab_k and exp_tab_k above reduce the SAME r twice, against two unrelated periods.  With kr > 0 and a decay rate that is small
// against it (|kappa| <= kr/4, kappa = -Im k: the physically usual case, a slightly absorbing medium) one reduction serves both:
//     r = n h + y,   h = (2 pi / kr) / 2048,  |y| <= h/2,   n = 2048 m + j   (m whole periods, j the node inside the period)
//     e^{(kappa + i kr) r} = D_m . T_j . e^{(kappa + i kr) y}
//     T_j = e^{kappa j h} (cos, sin)(2 pi j / 2048)     2048 complex nodes, the decay inside the period folded INTO the trig table
//     D_m = e^{kappa m 2 pi / kr}                       one real factor per whole period, kCexpPeriods of them (kr r < 1608)
//     e^{w} - 1, w = (kappa + i kr) y: |w| <= 1.031 pi/2048 = 1.58e-3, so degree 4 is exact to |w|^5/120 = 8.3e-17; y is REAL, so a Horner
//     step is two independent real FMAs (real and imaginary coefficient), 8 instructions in all — not the 4-FMA complex multiply-add that
//     DESIGN.md §4.1 costed in round 2.
// Per pair: 4 (reduction) + 8 (polynomial) + 4 (T_j (1 + P)) + 1 (D_m into the amplitude) = 17 fp64 instructions and two table reads,
// against 14 + 10 + 1 for the two separate functions; a real wavenumber needs no D_m and only the even / odd halves: 4 + 5 + 4 = 13
// against 14.  Tables: 2 x 2048 + 256 doubles = 34 KB of LDS per workgroup.
constexpr int kCexpShift = 11, kCexpNodes = 1 << kCexpShift, kCexpPeriods = 256;
constexpr int kCexpTableDoubles = 2 * kCexpNodes + kCexpPeriods;
constexpr double kCexpMaxPhase = 1600.0;          // kr r below kCexpPeriods whole periods (256 x 2 pi = 1608)
constexpr double kCexpMaxDecayRatio = 0.25;       // |kappa| <= kr / 4

struct CexpCoeffsK {
  double ih, h1, h2;                               // 1/h and -h in two pieces
  double a1r, a1i, a2r, a2i, a3r, a3i, a4r, a4i;   // (kappa + i kr)^m / m!
  static SCTL_AMD_HD bool usable(double kr, double kappa) { return kr > 0 && __builtin_fabs(kappa) <= kCexpMaxDecayRatio * kr && kr < 1e150 && kr > 1e-150; }
  SCTL_AMD_HD void set(double kr, double kappa, const TabCoeffs& B) { set_scaled(kr, kappa, B, 1.0); }
  // For a distance handed in as C r (the kernels' unnormalised reciprocal square root gives C / r, and r2 (C / r) = C r costs no more than r):
  // the constants of the wavenumber k / C.  The step C h keeps its two-piece accuracy — C (24 significant bits or a power of two) times the
  // 33-bit head is split exactly by one FMA —; the reciprocal step and the polynomial coefficients only need rounding accuracy.
  SCTL_AMD_HD void set_scaled(double kr_true, double kappa_true, const TabCoeffs& B, double C) {
    const double kr = kr_true / C, kappa = kappa_true / C;
    ih = (4 * B.inv_h) * kr;                       // 2048 / (2 pi) x kr / C
    double ph, pl;
    TabCoeffsK::mul2(C, B.h1 / 4, B.h2 / 4, ph, pl);
    TabCoeffsK::div2(ph, pl, kr_true, h1, h2);     // -(pi/1024) C / kr
    // powers of c = kappa + i kr
    const double c2r = kappa * kappa - kr * kr, c2i = 2 * kappa * kr;
    const double c3r = c2r * kappa - c2i * kr, c3i = c2r * kr + c2i * kappa;
    const double c4r = c3r * kappa - c3i * kr, c4i = c3r * kr + c3i * kappa;
    a1r = kappa; a1i = kr;
    a2r = 0.5 * c2r; a2i = 0.5 * c2i;
    a3r = B.p2 * c3r; a3i = B.p2 * c3i;            // 1/3!
    a4r = B.p3 * c4r; a4i = B.p3 * c4i;            // 1/4!
  }
#ifdef __HIPCC__
  __device__ __forceinline__ void pin() {
    TabCoeffsK::to_sgpr(ih); TabCoeffsK::to_sgpr(h1); TabCoeffsK::to_sgpr(h2);
    TabCoeffsK::to_sgpr(a1r); TabCoeffsK::to_sgpr(a1i); TabCoeffsK::to_sgpr(a2r); TabCoeffsK::to_sgpr(a2i);
    TabCoeffsK::to_sgpr(a3r); TabCoeffsK::to_sgpr(a3i); TabCoeffsK::to_sgpr(a4r); TabCoeffsK::to_sgpr(a4i);
  }
#endif
};

// ---- table fill in double-double arithmetic: every entry is the correctly rounded value (0.5 ulp), so the tables add nothing to the
// per-pair error beyond their own storage rounding.  One-time work per workgroup (8 entries per lane of 256), a fraction of a per cent of it.
// NO floating-point contraction in this block: hipcc contracts a * b + c into an FMA ACROSS statements by default (-ffp-contract=fast), which
// turns `p = a.h * b.h; ... s = p + e` into s = fma(a.h, b.h, e) and breaks the error-free transformations (measured on the device before
// this pragma: table entries off by j x 2.7e-17, i.e. 5.5e-14 at the last node, where the host build of the same code had 1e-16).
#if defined(__clang__)
#define SCTL_AMD_FP_EXACT _Pragma("clang fp contract(off)")
#else
#define SCTL_AMD_FP_EXACT          // g++ (host tests): built with -ffp-contract=off
#endif
struct DD { double h, l; };
SCTL_AMD_HD DD dd_norm(double h, double l) { SCTL_AMD_FP_EXACT const double s = h + l; return DD{s, l - (s - h)}; }
SCTL_AMD_HD DD dd_add(DD a, DD b) {
  SCTL_AMD_FP_EXACT
  const double s = a.h + b.h, bb = s - a.h;
  const double e = ((a.h - (s - bb)) + (b.h - bb)) + (a.l + b.l);
  return dd_norm(s, e);
}
SCTL_AMD_HD DD dd_neg(DD a) { return DD{-a.h, -a.l}; }
SCTL_AMD_HD DD dd_mul(DD a, DD b) {
  SCTL_AMD_FP_EXACT
  const double p = a.h * b.h;
  const double e = fma_(a.h, b.h, -p) + fma_(a.h, b.l, a.l * b.h);
  return dd_norm(p, e);
}
struct DDC { DD r, i; };
SCTL_AMD_HD DDC ddc_mul(DDC a, DDC b) { return DDC{dd_add(dd_mul(a.r, b.r), dd_neg(dd_mul(a.i, b.i))), dd_add(dd_mul(a.r, b.i), dd_mul(a.i, b.r))}; }
SCTL_AMD_HD DDC ddc_pow(DDC a, int n) {          // a^n, n >= 0, by squaring
  DDC r{DD{1, 0}, DD{0, 0}};
  while (n) {
    if (n & 1) r = ddc_mul(r, a);
    n >>= 1;
    if (n) a = ddc_mul(a, a);
  }
  return r;
}
SCTL_AMD_HD DD dd_pow(DD a, int n) {
  DD r{1, 0};
  while (n) {
    if (n & 1) r = dd_mul(r, a);
    n >>= 1;
    if (n) a = dd_mul(a, a);
  }
  return r;
}
// e^x for a small double-double x (|x| < 1e-3): Taylor to x^8 (next term < 3e-30)
SCTL_AMD_HD DD dd_exp_small(DD x) {
  const DD f[7] = {DD{2.48015873015873016e-05, 2.15119478667758816e-23}, DD{1.98412698412698413e-04, 1.72095582934207053e-22},
                   DD{1.38888888888888894e-03, -5.30054395437357706e-20}, DD{8.33333333333333322e-03, 1.15648231731787138e-19},
                   DD{4.16666666666666644e-02, 2.31296463463574266e-18}, DD{1.66666666666666657e-01, 9.25185853854297066e-18}, DD{0.5, 0}};   // 1/8! .. 1/2!
  DD p = f[0];
  for (int k = 1; k < 7; k++) p = dd_add(dd_mul(p, x), f[k]);
  p = dd_add(dd_mul(p, x), DD{1, 0});
  return dd_add(dd_mul(p, x), DD{1, 0});
}

// exp(kappa x) for x = i q with q = q1 + q2 given in two pieces (q1 with <= 33 significant bits, so i q1 is exact for i < 2^19): the product
// kappa (i q1) is split into its rounded value and the exact remainder, so that the result carries exp_fast's error only
// (used where the double-double power would leave the double range: the value is then 0, inf or about to be)
SCTL_AMD_HD double exp_of_multiple(int i, double q1, double q2, double kappa, const Coeffs& K) {
  SCTL_AMD_FP_EXACT
  const double x1 = i * q1;
  const double p = kappa * x1;
  const double lo = fma_(kappa, x1, -p) + kappa * (i * q2);
  const double e = exp_fast(p, K);
  return fma_(e, fma_(lo, 0.5 * lo, lo), e);         // e (1 + lo + lo^2 / 2): |lo| < 1e-7
}
// the 2048 complex nodes T_j and the period factors D_m, filled cooperatively (lane `tid` of `nthreads`); kr, kappa as in CexpCoeffsK::set.
// T_1 = e^{kappa h} (cos + i sin)(pi/1024) in double-double (kr h = pi/1024 exactly: the phase step does not depend on k), T_j = T_1^j;
// D_1 = (e^{kappa h})^2048, D_m = D_1^m.
SCTL_AMD_HD void fill_cexp_tables(double* table, int tid, int nthreads, double kr, double kappa, const Coeffs& K, const TabCoeffs& B) {
  double h1, h2;
  TabCoeffsK::div2(-B.h1 / 4, -B.h2 / 4, kr, h1, h2);          // +h in two pieces
  const DD x = dd_mul(DD{kappa, 0}, dd_norm(h1, h2));          // kappa h, |x| <= pi/4096
  const DD eh = dd_exp_small(x);
  const DDC c1{DD{9.99995293809576191e-01, -1.96680642853221887e-17}, DD{3.06795676296597614e-03, 1.26902790854559250e-19}};   // e^{i pi/1024}
  const DDC t1{dd_mul(eh, c1.r), dd_mul(eh, c1.i)};
  if (tid < kCexpNodes) {
    DDC t = ddc_pow(t1, tid);
    const DDC step = ddc_pow(t1, nthreads);
    for (int j = tid; j < kCexpNodes; j += nthreads) {
      table[2 * j] = t.r.h;
      table[2 * j + 1] = t.i.h;
      if (j + nthreads < kCexpNodes) t = ddc_mul(t, step);
    }
  }
  // whole periods: |kappa m 2 pi / kr| stays below ~690 for the double-double power; beyond, the plain exponential (0 / inf in the limit)
  long long hb;
  __builtin_memcpy(&hb, &h1, 8);
  hb &= ~((1ll << 20) - 1);                                    // hh: the leading 33 bits of h1, so that (2048 m) hh (< 2^19 hh) is exact
  double hh;
  __builtin_memcpy(&hh, &hb, 8);
  const double hl = (h1 - hh) + h2;
  const DD d1 = dd_pow(eh, kCexpNodes);
  for (int m = tid; m < kCexpPeriods; m += nthreads) {
    const double est = __builtin_fabs(kappa * (m * (double)kCexpNodes * h1));
    table[2 * kCexpNodes + m] = (kappa == 0 || m == 0) ? 1.0 : (est < 690.0 ? dd_pow(d1, m).h : exp_of_multiple(m * kCexpNodes, hh, hl, kappa, K));
  }
}

// (re, im) of T_j e^{w} and the period factor D_m for a distance r; precondition kr r <= kCexpMaxPhase (indices are masked, so an argument
// beyond it — or a NaN — reads inside the table and returns garbage that the caller's range check discards)
SCTL_AMD_HD void cexp_tab_k(double r, double& re, double& im, double& dm, const CexpCoeffsK& K, const double* table) {
  const double t = fma_(r, K.ih, kRoundMagic);
  const double n = t - kRoundMagic;
  double y = fma_(n, K.h1, r);
  y = fma_(n, K.h2, y);
  const int ni = low_dword(t);
  const int j = ni & (kCexpNodes - 1);
  const double tr = table[2 * j], ti = table[2 * j + 1];
  dm = table[2 * kCexpNodes + ((ni >> kCexpShift) & (kCexpPeriods - 1))];
  double pr = fma_(K.a4r, y, K.a3r), pi = fma_(K.a4i, y, K.a3i);
  pr = fma_(pr, y, K.a2r); pi = fma_(pi, y, K.a2i);
  pr = fma_(pr, y, K.a1r); pi = fma_(pi, y, K.a1i);
  pr *= y; pi *= y;                                  // e^w - 1
  re = fma_(tr, pr, fma_(-ti, pi, tr));
  im = fma_(ti, pr, fma_(tr, pi, ti));
}
// real wavenumber (kappa = 0): e^{i kr y} - 1 = z (a2r + a4r z) + i y (a1i + a3i z), z = y^2; no period factor
SCTL_AMD_HD void cexp_tab_k_real(double r, double& re, double& im, const CexpCoeffsK& K, const double* table) {
  const double t = fma_(r, K.ih, kRoundMagic);
  const double n = t - kRoundMagic;
  double y = fma_(n, K.h1, r);
  y = fma_(n, K.h2, y);
  const int j = low_dword(t) & (kCexpNodes - 1);
  const double tr = table[2 * j], ti = table[2 * j + 1];
  const double z = y * y;
  const double pr = z * fma_(z, K.a4r, K.a2r);
  const double pi = y * fma_(z, K.a3i, K.a1i);
  re = fma_(tr, pr, fma_(-ti, pi, tr));
  im = fma_(ti, pr, fma_(tr, pi, ti));
}

SCTL_AMD_HD double exp_tab(double x, const TabCoeffs& K, const double* table) {
  const double e = exp_tab_clamped(__builtin_fmin(__builtin_fmax(x, -800.0), 800.0), K, table);
  return (x != x) ? x : e;                  // fmin/fmax drop a NaN: restore it
}

}  // namespace fastmath
}  // namespace sctl_amd
