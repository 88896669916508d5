// CPU check of sctl_amd/csrc/fastmath.hpp (the fp64 sincos / exp used by the Helmholtz device kernel) against libm in
// long double (__float128 where a product of two doubles must be exact).  Prints the maximum errors; tests/test_fastmath.py asserts the
// bounds.  Built with g++ -lquadmath (no HIP needed).
#include "../../include/sctl_amd/device/fastmath.hpp"

#include <cmath>
#include <quadmath.h>
#include <cstdio>
#include <cstdlib>

int main() {
  using namespace sctl_amd::fastmath;
  const Coeffs K;
  srand48(1);
  const TabCoeffs T;
  static double table[kTableDoubles];
  for (int t = 0; t < 7; t++) fill_tables(table, t, 7, K);      // any cooperative split fills every entry
  double es = 0, ec = 0, ee = 0, ts = 0, tc = 0, te = 0;
  for (int i = 0; i < 4000000; i++) {
    const double scale = (i % 5 == 0) ? 1.0 : (i % 5 == 1) ? 30.0 : (i % 5 == 2) ? 3000.0 : (i % 5 == 3) ? 1.2e4 : 1.5e6;
    const double x = (drand48() * 2 - 1) * scale;
    double s, c;
    sincos_reduced(x, s, c, K);
    const long double sl = sinl((long double)x), cl = cosl((long double)x);
    es = fmax(es, fabs((double)(s - sl)));
    ec = fmax(ec, fabs((double)(c - cl)));
    const double t = (drand48() * 2 - 1) * ((i % 2) ? 5.0 : 700.0);
    const long double el = expl((long double)t);
    ee = fmax(ee, fabs((double)((exp_fast(t, K) - el) / el)));
    te = fmax(te, fabs((double)((exp_tab(t, T, table) - el) / el)));
    if (fabs(x) <= kSincosTabMaxArg) {
      sincos_tab(x, s, c, T, table);
      ts = fmax(ts, fabs((double)(s - sl)));
      tc = fmax(tc, fabs((double)(c - cl)));
    }
  }
  // the forms with the wavenumber folded in: sincos(kr r), exp(kappa r) against long double of the EXACT product
  double ks = 0, kc = 0, ke = 0;
  bool k_specials = true;
  const double krs[6] = {7.5, 0.013, 250.0, -3.2, 1e-9, 0.0}, kappas[6] = {-0.3, 0.3, -25.0, 1e-3, 40.0, 0.0};
  for (int m = 0; m < 6; m++) {
    TabCoeffsK TK;
    TK.set(krs[m], kappas[m], T);
    const double rmax_s = (krs[m] != 0) ? fmin(kSincosTabMaxArg / fabs(krs[m]), 1e6) : 1e6;
    const double rmax_e = (kappas[m] != 0) ? fmin(700.0 / fabs(kappas[m]), 1e6) : 1e6;   // relative accuracy where the result is a normal number
    for (int i = 0; i < 400000; i++) {
      const double u = drand48(), r = ((i % 3 == 0) ? u * u * u : u) * rmax_s;
      double s, c;
      sincos_tab_k(r, s, c, TK, table);
      const __float128 x = (__float128)krs[m] * (__float128)r;      // the product of two doubles is exact in 113 bits (not in long double's 64)
      ks = fmax(ks, fabs((double)((__float128)s - sinq(x))));
      kc = fmax(kc, fabs((double)((__float128)c - cosq(x))));
      const double re = ((i % 3 == 0) ? u * u * u : u) * rmax_e;
      const __float128 el = expq((__float128)kappas[m] * (__float128)re);
      ke = fmax(ke, fabs((double)(((__float128)exp_tab_k(re, TK, table) - el) / el)));
    }
    double s, c;
    sincos_tab_k(0.0, s, c, TK, table);
    k_specials = k_specials && s == 0.0 && c == 1.0 && exp_tab_k(0.0, TK, table) == 1.0;
    if (krs[m] == 0) { sincos_tab_k(123.456, s, c, TK, table); k_specials = k_specials && s == 0.0 && c == 1.0; }
    if (kappas[m] == 0) k_specials = k_specials && exp_tab_k(123.456, TK, table) == 1.0;
    else {   // beyond the double range, up to the stated limit of the argument: 0 or inf; NaN in -> NaN out
      const double far = 0.9 * kExpTabMaxArg / fabs(kappas[m]), v = exp_tab_k(far, TK, table), w = exp_tab_k(NAN, TK, table);
      k_specials = k_specials && (kappas[m] < 0 ? v == 0.0 : std::isinf(v)) && w != w;
    }
  }
  printf("k_abs_err_sin %.3e\nk_abs_err_cos %.3e\nk_rel_err_exp %.3e\nk_specials %d\n", ks, kc, ke, (int)k_specials);
  // ONE reduction for the whole factor e^{(kappa + i kr) r} (cexp_tab_k): error of D_m (re + i im) relative to |e^{(kappa + i kr) r}|, against
  // 113-bit arithmetic of the exact products; wavenumbers at both ends of the allowed decay ratio, tiny and large kr, kappa of either sign
  {
    static double ctab[kCexpTableDoubles];
    const double ckr[7] = {7.5, 7.5, 0.013, 250.0, 3.2, 1e-9, 40.0}, ckap[7] = {-0.3, 0.0, 0.013 / 4, -62.5, 0.8, 0.0, -1e-3};
    double ce = 0, cre = 0, two = 0;
    bool c_specials = true;
    for (int m = 0; m < 7; m++) {
      c_specials = c_specials && CexpCoeffsK::usable(ckr[m], ckap[m]);
      CexpCoeffsK CK;
      CK.set(ckr[m], ckap[m], T);
      TabCoeffsK TK;                       // the two-reduction form on the same arguments, for comparison
      TK.set(ckr[m], ckap[m], T);
      for (int t = 0; t < 5; t++) fill_cexp_tables(ctab, t, 5, ckr[m], ckap[m], K, T);
      const double rmax = fmin(kCexpMaxPhase / ckr[m], 1e6);
      for (int i = 0; i < 400000; i++) {
        const double u = drand48(), r = ((i % 3 == 0) ? u * u * u : u) * rmax;
        double re, im, dm;
        cexp_tab_k(r, re, im, dm, CK, ctab);
        const __float128 ph = (__float128)ckr[m] * (__float128)r, mag = expq((__float128)ckap[m] * (__float128)r);
        const __float128 er = (__float128)dm * (__float128)re - mag * cosq(ph), ei = (__float128)dm * (__float128)im - mag * sinq(ph);
        ce = fmax(ce, (double)(sqrtq(er * er + ei * ei) / mag));
        {
          double s2, c2;
          sincos_tab_k(r, s2, c2, TK, table);
          const double a2 = exp_tab_k(r, TK, table);
          const __float128 fr = (__float128)(a2 * c2) - mag * cosq(ph), fi = (__float128)(a2 * s2) - mag * sinq(ph);
          two = fmax(two, (double)(sqrtq(fr * fr + fi * fi) / mag));
        }
        if (ckap[m] == 0) {
          cexp_tab_k_real(r, re, im, CK, ctab);
          const __float128 fr = (__float128)re - cosq(ph), fi = (__float128)im - sinq(ph);
          cre = fmax(cre, (double)sqrtq(fr * fr + fi * fi));
        }
      }
      double re, im, dm;
      cexp_tab_k(0.0, re, im, dm, CK, ctab);
      c_specials = c_specials && re == 1.0 && im == 0.0 && dm == 1.0;
    }
    c_specials = c_specials && !CexpCoeffsK::usable(0.0, 0.0) && !CexpCoeffsK::usable(-7.5, 0.1) && !CexpCoeffsK::usable(1.0, 0.26) && !CexpCoeffsK::usable(1.0, -0.26);
    printf("cexp_rel_err %.3e\ncexp_real_abs_err %.3e\ncexp_specials %d\ntwo_reductions_rel_err %.3e\n", ce, cre, (int)c_specials, two);
  }
  double s1, c1;
  sincos_tab(0.0, s1, c1, T, table);
  const bool tab_specials = exp_tab(-1e9, T, table) == 0.0 && std::isinf(exp_tab(1e9, T, table)) && exp_tab(0.0, T, table) == 1.0 &&
                            exp_tab(NAN, T, table) != exp_tab(NAN, T, table) && s1 == 0.0 && c1 == 1.0;
  printf("tab_abs_err_sin %.3e\ntab_abs_err_cos %.3e\ntab_rel_err_exp %.3e\ntab_specials %d\n", ts, tc, te, (int)tab_specials);
  // specials
  const double z = exp_fast(-1e9, K), big = exp_fast(1e9, K), one = exp_fast(0.0, K), nn = exp_fast(NAN, K);
  double s0, c0;
  sincos_reduced(0.0, s0, c0, K);
  printf("max_abs_err_sin %.3e\nmax_abs_err_cos %.3e\nmax_rel_err_exp %.3e\n", es, ec, ee);
  printf("specials %d\n", (z == 0.0) && std::isinf(big) && (one == 1.0) && (nn != nn) && (s0 == 0.0) && (c0 == 1.0));
  return 0;
}
