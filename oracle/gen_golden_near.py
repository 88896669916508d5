#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — golden data for the BoundaryIntegralOp NEAR field, generated from the REAL reference
(oracle/_ref/libsctl_ref_near.so: the reference's BoundaryIntegralOp driven with the synthetic PatchElemList of
oracle/ref_near_shim.cpp).  Build container only.  Per case tests/golden/near_field.npz holds
  * the near-operator arrays the reference's SetupNear produced (boundary_integral.txx:816-1012): they are the INPUT of
    sctl_amd_near_create, and what include/sctl_amd/boundary_integral.hpp's own SetupNear must reproduce,
  * u_near = ComputeNearInterac(F) and u_total = ComputePotential(F) of the reference: the expected outputs.
Point coordinates, normals, weights and densities are NOT stored: near_inputs() regenerates them from the seed.

    python oracle/gen_golden_near.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sctl_amd.rand48 import Rand48  # noqa: E402

# kernel, Nt (0: targets = surface nodes), Ns, nodes per element, far-field upsampling, dot with target normals, far-field distance
CASES = [
    ("Laplace3D-FxU", 300, 200, 4, 1, 0, 0.15),
    ("Laplace3D-FxU", 0, 203, 4, 2, 0, 0.15),
    ("Laplace3D-DxU", 0, 150, 6, 1, 0, 0.20),
    ("Laplace3D-FxdU", 120, 100, 5, 1, 1, 0.20),
    ("Stokes3D-FxU", 0, 96, 4, 1, 0, 0.20),
    ("Stokes3D-DxU", 150, 120, 5, 1, 0, 0.20),
    ("Stokes3D-FxT", 100, 90, 3, 2, 1, 0.20),
]
# two element lists in one operator, the second one matrix-free with a near zone: kernel, Nt, Ns (all nodes), nodes of the
# matrix-free list (the LAST ones), nodes per element, upsampling, dot, far-field distance
CASES2 = [
    ("Laplace3D-FxU", 250, 230, 90, 4, 1, 0, 0.18),
    ("Stokes3D-DxU", 0, 150, 61, 5, 2, 0, 0.20),
    ("Laplace3D-FxdU", 140, 120, 50, 3, 1, 1, 0.20),
]


def near_inputs(seed, Nt, Ns, k0):
    """drand48 draw order shared with tests/cpp/bie_driver.cpp: targets, target normals, nodes, node normals, weights, density."""
    g = Rand48(seed)
    xt = g.drand48(Nt * 3) - 0.5
    xnt = g.drand48(Nt * 3) - 0.5
    xs = g.drand48(Ns * 3) - 0.5
    xn = g.drand48(Ns * 3) - 0.5
    w = g.drand48(Ns) * 0.01
    f = g.drand48(Ns * k0) - 0.5
    return xt, xnt, xs, xn, w, f


def main():
    O = oracle.restatement()
    arrays, cases = {}, []
    for i, (name, Nt, Ns, npe, ups, dot, rad) in enumerate(CASES):
        inf = O.info(name)
        seed = 700 + i
        xt, xnt, xs, xn, w, f = near_inputs(seed, Nt, Ns, inf["k0"])
        r = oracle.reference_near(name, xt if Nt else None, xnt if Nt else None, xs, xn, w, f, bool(dot), 1e-10, npe, ups, rad)
        key = "c%d" % i
        for k, v in r.items():
            arrays["%s/%s" % (key, k)] = v
        cases.append(dict(key=key, kernel=name, Nt=Nt, Ns=Ns, seed=seed, nodes_per_elem=npe, upsample=ups, trg_normal_dot_prod=dot, rad=rad,
                          near_entries=int(r["near_scatter_index"].size), K_near_len=int(r["K_near"].size)))
        print(name, cases[-1])
    for i, (name, Nt, Ns, nfree, npe, ups, dot, rad) in enumerate(CASES2):
        inf = O.info(name)
        seed = 800 + i
        xt, xnt, xs, xn, w, f = near_inputs(seed, Nt, Ns, inf["k0"])
        r = oracle.reference_near(name, xt if Nt else None, xnt if Nt else None, xs, xn, w, f, bool(dot), 1e-10, npe, ups, rad, free_nodes=nfree)
        key = "t%d" % i
        for k, v in r.items():
            arrays["%s/%s" % (key, k)] = v
        cases.append(dict(key=key, kernel=name, Nt=Nt, Ns=Ns, seed=seed, nodes_per_elem=npe, upsample=ups, trg_normal_dot_prod=dot, rad=rad, free_nodes=nfree,
                          near_entries=int(r["near_scatter_index"].size), K_near_len=int(r["K_near"].size)))
        print(name, cases[-1])
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "near_field.npz"), **arrays)
    with open(os.path.join(out, "near_manifest.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden_near.py", "cases": cases}, fh, indent=1)
    print("near_field.npz: %.1f KB" % (os.path.getsize(os.path.join(out, "near_field.npz")) / 1024))


if __name__ == "__main__":
    main()
