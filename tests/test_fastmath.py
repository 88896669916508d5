"""CPU check of the fp64 sincos / exp the Helmholtz device kernel uses (include/sctl_amd/device/fastmath.hpp is host+device code)."""
import os
import re
import subprocess

from conftest import ROOT


def test_fastmath_against_libm(tmp_path):
    exe = str(tmp_path / "fastmath_check")
    subprocess.run(["g++", "-O2", "-std=c++14", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "fastmath_check.cpp"), "-o", exe, "-lquadmath"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    v = {k: float(x) for k, x in re.findall(r"(\w+) ([0-9.e+-]+)", out)}
    assert v["max_abs_err_sin"] < 4e-16 and v["max_abs_err_cos"] < 4e-16      # |x| up to 1.5e6
    assert v["max_rel_err_exp"] < 4e-16                                        # |x| up to 700
    assert v["specials"] == 1                                                  # exp(-huge)=0, exp(huge)=inf, exp(0)=1, NaN, sincos(0)
    # the table-driven forms the device loop runs (tables filled by the polynomial code, as a workgroup does)
    assert v["tab_abs_err_sin"] < 4e-16 and v["tab_abs_err_cos"] < 4e-16      # |x| up to 1.2e4
    assert v["tab_rel_err_exp"] < 4e-16 and v["tab_specials"] == 1
    # the same with the launch's wavenumber folded into the reduction (what the speculative pass runs): six (Re k, -Im k) pairs incl. 0
    assert v["k_abs_err_sin"] < 4e-16 and v["k_abs_err_cos"] < 4e-16 and v["k_rel_err_exp"] < 4e-16 and v["k_specials"] == 1
    # round 3: ONE reduction of r for the whole factor e^{ikr} (2048 complex nodes with the decay folded in, degree-4 complex polynomial in a
    # real argument): |error| / |e^{ikr}| over seven wavenumbers up to the allowed decay ratio |Im k| = Re k / 4, phases up to 1600
    # (tables filled in double-double arithmetic: correctly rounded entries).  The measure is the MODULUS of the complex error, which for the
    # two-reduction form — the product of a 2.5e-16 exponential and a 2.9e-16 sine / cosine — is 4.9e-16 on the same arguments: the new
    # form must not be worse than the one it replaces, and a real wavenumber stays below 3e-16.
    assert v["cexp_rel_err"] < 5e-16 and v["cexp_rel_err"] <= v["two_reductions_rel_err"] and v["cexp_real_abs_err"] < 3e-16 and v["cexp_specials"] == 1
