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
