#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — reference-made fixtures at the sizes of BASELINE.json's five configs.

Runs in the build container only (needs oracle/_ref/, i.e. /root/reference compiled by oracle/Makefile).
As in gen_golden.py the inputs are NOT stored: they are regenerated from sctl_amd.rand48 (bit-exact drand48, the
generator the reference's own driver draws from, fmm-wrapper.txx:41-55) in the driver's order targets, sources,
normals, densities; points in [0,1)^3 as bench.py's clouds.  Stored: outputs of the REAL reference.

  config 1 (Laplace3D-FxU, 16384 x 16384, fp64 — the size the reference's CPU path is quoted on, src/test-fmm.cpp:6 /
            fmm-wrapper.txx:35-92 style driver): GenericKernel::Eval at digits -1 and 10, and ParticleFMM::EvalDirect at
            accuracy 10 (fmm-wrapper.txx:58), ALL 16384 targets.
  configs 2-5 (+ Laplace FxU / DxU at 2^20): the reference's GenericKernel::Eval for a fixed 512-target subset
            (indices offset + i*stride) against ALL N seeded sources.  For the fp32 config both the reference's fp32 result
            and its fp64 result on the same (fp32-rounded) inputs are kept: the fp32 bound is stated against fp64.

    python oracle/gen_golden_fullsize.py        # rewrites tests/golden/fullsize.npz and tests/golden/fullsize_manifest.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sctl_amd.rand48 import point_cloud  # noqa: E402

HELMHOLTZ_K = [7.5, 0.3]
NSEL = 512
# (key, kernel, N, dtype, [digits...], BASELINE.json config index or None)
SUBSET_CASES = [
    ("laplace_sl_1m", "Laplace3D-FxU", 1 << 20, "f64", (-1, 10), None),
    ("laplace_dl_1m", "Laplace3D-DxU", 1 << 20, "f64", (-1,), None),
    ("cfg2_laplace_sldl_1m", "Laplace3D-FDxUdU", 1 << 20, "f64", (-1, 10), 1),
    ("cfg3_stokeslet_256k", "Stokes3D-FxU", 1 << 18, "f64", (-1, 10), 2),
    ("cfg5_helmholtz_1m", "Helmholtz3D-FxU", 1 << 20, "f64", (-1,), 4),
    ("cfg4_laplace_sl_f32_8m", "Laplace3D-FxU", 1 << 23, "f32", (-1,), 3),
]
CFG1 = ("cfg1_laplace_sl_16k", "Laplace3D-FxU", 1 << 14, "f64", 0)


def subset(N, nsel=NSEL):
    stride = N // nsel
    return stride // 2, stride          # offset, stride


def fullsize_inputs(seed, N, k0, nd, dtype):
    """Shared with tests/: (r_trg, r_src, n_src, v_src), points in [0,1)^3."""
    return point_cloud(seed, N, N, k0, nd, dtype, shift=0.0)


def main():
    R = oracle.reference()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    out_dir = os.path.join(ROOT, "tests", "golden")
    arrays, cases = {}, []
    seed = 9000

    key, name, N, tag, cfg = CFG1
    seed += 1
    inf = R.info(name)
    xt, xs, xn, f = fullsize_inputs(seed, N, inf["k0"], inf["nd"], np.float64)
    for digits in (-1, 10):
        arrays["%s_d%d" % (key, digits)] = R.eval(name, xt, xs, xn, f, digits=digits)
        cases.append(dict(key="%s_d%d" % (key, digits), kind="eval_all", kernel=name, N=N, dtype=tag, seed=seed, digits=digits, config=cfg))
    arrays[key + "_fmm10"] = R.particle_fmm_eval_direct(name, xt, xs, xn, f, digits=10)
    cases.append(dict(key=key + "_fmm10", kind="particle_fmm_all", kernel=name, N=N, dtype=tag, seed=seed, digits=10, config=cfg))
    print(key, "ParticleFMM vs Eval(10):", oracle.rel_l2(arrays[key + "_fmm10"], arrays[key + "_d10"]))

    for key, name, N, tag, digit_list, cfg in SUBSET_CASES:
        seed += 1
        inf = R.info(name)
        dt = np.float64 if tag == "f64" else np.float32
        ctx = np.array(HELMHOLTZ_K) if name.startswith("Helmholtz") else None
        t0 = time.time()
        xt, xs, xn, f = fullsize_inputs(seed, N, inf["k0"], inf["nd"], dt)
        off, stride = subset(N)
        sel = off + stride * np.arange(NSEL)
        xt_sel = xt.reshape(N, 3)[sel].ravel().copy()
        for digits in digit_list:
            k = "%s_d%d" % (key, digits)
            arrays[k] = R.eval(name, xt_sel, xs, xn, f, ctx=ctx, digits=digits)
            cases.append(dict(key=k, kind="eval_subset", kernel=name, N=N, dtype=tag, seed=seed, digits=digits, config=cfg,
                              nsel=NSEL, sel_offset=int(off), sel_stride=int(stride)))
        if tag == "f32":
            k = key + "_as_f64"
            arrays[k] = R.eval(name, *[a.astype(np.float64) for a in (xt_sel, xs, xn, f)], ctx=ctx, digits=-1)
            cases.append(dict(key=k, kind="eval_subset_f64_of_f32_inputs", kernel=name, N=N, dtype="f64", seed=seed, digits=-1, config=cfg,
                              nsel=NSEL, sel_offset=int(off), sel_stride=int(stride)))
            print(key, "reference fp32 vs its own fp64 on the same inputs:", oracle.rel_l2(arrays[key + "_d-1"], arrays[k]))
        print(key, "%.1f s" % (time.time() - t0))

    np.savez_compressed(os.path.join(out_dir, "fullsize.npz"), **arrays)
    with open(os.path.join(out_dir, "fullsize_manifest.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden_fullsize.py", "reference_isa": R.isa, "helmholtz_k": HELMHOLTZ_K,
                   "inputs": "sctl_amd.rand48.point_cloud(seed, N, N, K0, ND, dtype, shift=0.0)", "cases": cases}, fh, indent=1)
    print("tests/golden/fullsize.npz: %.1f KB" % (os.path.getsize(os.path.join(out_dir, "fullsize.npz")) / 1024))


if __name__ == "__main__":
    main()
