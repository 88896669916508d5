#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — golden data for the user-plugin test (tests/test_plugin.py): the Yukawa functor of
oracle/ref_shim.cpp (ref_ext::Yukawa3D_FxU) on the REAL reference's GenericKernel.  Build container only.
Inputs are regenerated from the seeds (sctl_amd.rand48.point_cloud); outputs go to tests/golden/Yukawa3D-FxU.npz.

    python oracle/gen_golden_plugin.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sctl_amd.rand48 import point_cloud  # noqa: E402

NAME, LAMBDA = "Yukawa3D-FxU", 2.5
CASES = [("f64", 257, 1000, -1, False), ("f64", 1024, 1024, -1, True), ("f32", 500, 700, -1, False), ("f64", 300, 400, 10, False), ("f64", 64, 33, -1, False)]


def main():
    R = oracle.reference()
    assert R is not None and R.info(NAME)["k0"] == 1
    arrays, cases = {}, []
    for i, (dt, Nt, Ns, digits, self_trg) in enumerate(CASES):
        seed = 950 + i
        xt, xs, xn, f = point_cloud(seed, Nt, Ns, 1, 0, np.float64 if dt == "f64" else np.float32)
        if self_trg:
            xt = xs
        key = "y%d" % i
        arrays[key] = R.eval(NAME, xt, xs, xn, f, ctx=np.array([LAMBDA]), digits=digits)
        cases.append(dict(key=key, dtype=dt, Nt=Nt, Ns=Ns, digits=digits, self_targets=int(self_trg), seed=seed))
    xt, xs, xn, f = point_cloud(960, 40, 30, 1, 0, np.float64)
    arrays["matrix"] = R.kernel_matrix(NAME, xt, xs, xn, ctx=np.array([LAMBDA]))
    arrays["manifest"] = np.frombuffer(json.dumps(dict(name=NAME, lam=LAMBDA, cases=cases, matrix=dict(seed=960, Nt=40, Ns=30))).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", NAME + ".npz"), **arrays)
    print(cases)


if __name__ == "__main__":
    main()
