#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — golden data for the batched list evaluation (sctl_amd_lists_*): the REAL
reference's GenericKernel::Eval (the type-erased entry PVFMM's P2P wrapper reaches, fmm-wrapper.txx:756-786), called once per
(target box, source box) list on sub-ranges of the particle arrays and accumulated — exactly what one device launch replaces.
Build container only.  Inputs are regenerated from the seed by tests/test_lists.py (list_case_inputs below); only outputs are stored.

    python oracle/gen_golden_lists.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sctl_amd.lists import grid_neighbour_lists, points_in_boxes  # noqa: E402

# kernel, dtype, grid (boxes per dimension), max points per box, targets are the sources (self interaction), digits
CASES = [
    ("Laplace3D-FxU", "f64", 8, 500, False, -1),     # 10 648 lists of 1-500 points: the size VERDICT r1 item 5 names
    ("Laplace3D-FxU", "f64", 5, 300, True, -1),      # every box also acts on itself with coincident points
    ("Stokes3D-DxU", "f64", 5, 120, False, -1),      # normals, 3x3
    ("Laplace3D-FxdU", "f32", 5, 200, False, -1),
    ("Helmholtz3D-FxU", "f64", 4, 150, True, -1),    # context blob, self interaction
    ("Stokes3D-FxU", "f64", 4, 100, False, 10),      # the 10 digits ParticleFMM asks for
]
HELMHOLTZ_K = [7.5, 0.3]


def list_case_inputs(seed, grid, max_pts, self_trg, k0, nd, dtype):
    """Per-box counts uniform in [1, max_pts], points inside their boxes, normals and densities uniform in [-0.5, 0.5)."""
    rng = np.random.default_rng(seed)
    nb = grid ** 3
    cs = rng.integers(1, max_pts + 1, nb)
    ct = cs if self_trg else rng.integers(1, max_pts + 1, nb)
    xs = points_in_boxes(grid, cs, rng, dtype)
    xt = xs if self_trg else points_in_boxes(grid, ct, rng, dtype)
    ns = int(cs.sum())
    xn = (rng.random(ns * nd) - 0.5).astype(dtype)
    f = (rng.random(ns * k0) - 0.5).astype(dtype)
    return ct, cs, xt, xs, xn, f


def main():
    R = oracle.reference()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    arrays, cases = {}, []
    for i, (name, dt, grid, max_pts, self_trg, digits) in enumerate(CASES):
        inf = R.info(name)
        npdt = np.float64 if dt == "f64" else np.float32
        seed = 900 + i
        ct, cs, xt, xs, xn, f = list_case_inputs(seed, grid, max_pts, self_trg, inf["k0"], inf["nd"], npdt)
        to, tc, so, sc = grid_neighbour_lists(grid, ct, cs)
        ctx = np.array(HELMHOLTZ_K) if name.startswith("Helmholtz") else None
        k0, k1, nd = inf["k0"], inf["k1"], inf["nd"]
        u = np.zeros(int(ct.sum()) * k1, dtype=npdt)
        for l in range(to.size):          # one reference Eval per list, accumulating into the target range (generic-kernel.txx:182-186)
            t0, t1, s0, s1 = int(to[l]), int(to[l] + tc[l]), int(so[l]), int(so[l] + sc[l])
            ut = u[t0 * k1:t1 * k1]
            R.eval(name, xt[t0 * 3:t1 * 3].copy(), xs[s0 * 3:s1 * 3].copy(), xn[s0 * nd:s1 * nd].copy(), f[s0 * k0:s1 * k0].copy(), v_trg=ut, ctx=ctx, digits=digits, omp=False)
        key = "l%d" % i
        arrays[key] = u
        cases.append(dict(key=key, kernel=name, dtype=dt, grid=grid, max_pts=max_pts, self_targets=int(self_trg), digits=digits, seed=seed, nlists=int(to.size),
                          Nt=int(ct.sum()), Ns=int(cs.sum()), pairs=int((tc * sc).sum())))
        print(cases[-1])
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "p2p_lists.npz"), **arrays)
    with open(os.path.join(out, "lists_manifest.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden_lists.py", "helmholtz_k": HELMHOLTZ_K, "cases": cases}, fh, indent=1)
    print("p2p_lists.npz: %.1f KB" % (os.path.getsize(os.path.join(out, "p2p_lists.npz")) / 1024))


if __name__ == "__main__":
    main()
