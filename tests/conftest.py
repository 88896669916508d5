import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
HELMHOLTZ_K = [7.5, 0.3]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_terminal_summary(terminalreporter):
    """Say WHY a test was skipped in the last lines of the run (the driver records only the tail of `pytest -q`): on a one-GPU box the
    RCCL tests that need one GPU per rank are skipped, and the record should show that rather than a bare `1 skipped`."""
    for rep in terminalreporter.stats.get("skipped", []):
        why = rep.longrepr[2] if isinstance(rep.longrepr, tuple) and len(rep.longrepr) == 3 else str(rep.longrepr)
        terminalreporter.write_line("SKIPPED %s -- %s" % (rep.nodeid, why))


def ctx_for(name):
    return np.array(HELMHOLTZ_K) if name.startswith("Helmholtz") else None


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


_NPZ = {}


def golden_array(kernel, key):
    if kernel not in _NPZ:
        _NPZ[kernel] = np.load(os.path.join(GOLDEN, kernel + ".npz"))
    return _NPZ[kernel][key]


def case_inputs(case, info):
    """Regenerate the inputs of a golden case from its seed (sctl_amd.rand48 == POSIX drand48)."""
    from sctl_amd.rand48 import Rand48, point_cloud
    dt = np.float64 if case["dtype"] == "f64" else np.float32
    xt, xs, xn, f = point_cloud(case["seed"], case["Nt"], case["Ns"], info["k0"], info["nd"], dt)
    if case["kind"] == "eval_self":
        xt = xs
    v0 = None
    if case["kind"] == "eval_accumulate":
        v0 = (Rand48(case["prefill_seed"]).drand48(case["Nt"] * info["k1"]) - 0.5).astype(dt)
    return xt, xs, xn, f, v0


def load_fullsize_manifest():
    with open(os.path.join(GOLDEN, "fullsize_manifest.json")) as fh:
        return json.load(fh)


_FULL_INPUTS = {}


def fullsize_inputs(case, info):
    """Inputs of a full-size golden case (oracle/gen_golden_fullsize.py): drand48 in the reference driver's order, points in
    [0,1)^3; the subset's target indices.  Cached per (seed, N, dtype): several cases share one cloud."""
    from sctl_amd.rand48 import point_cloud
    dt = np.float32 if case["key"].startswith("cfg4") and case["kind"] != "eval_all" else np.float64
    key = (case["seed"], case["N"], dt)
    if key not in _FULL_INPUTS:
        _FULL_INPUTS.clear()                     # one cloud at a time: the 2^23 case is 235 MB
        _FULL_INPUTS[key] = point_cloud(case["seed"], case["N"], case["N"], info["k0"], info["nd"], dt, shift=0.0)
    sel = None
    if "nsel" in case:
        sel = case["sel_offset"] + case["sel_stride"] * np.arange(case["nsel"])
    return _FULL_INPUTS[key] + (sel,)


def fullsize_array(key):
    if "fullsize" not in _NPZ:
        _NPZ["fullsize"] = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    return _NPZ["fullsize"][key]


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def tol_for(case):
    """Tolerance (rel-L2) of a result against the reference's stored output.
    f64 full precision: 1e-12 (BASELINE.json north_star; both sides sit at ~1e-15).
    f32: 2e-5 — the reference's own f32 result is 2e-6..8e-5 from the long-double truth at N=4096,
    summation order differs between the AVX-512 path and the GPU tiles.
    digits=d: the reference's approximate rsqrt is only promised to d digits; kernels use up to rinv^5, so 10*10^-d."""
    if case["digits"] >= 0 and case["digits"] < 15:
        return 10.0 * 10.0 ** (-case["digits"])
    return 1e-12 if case["dtype"] == "f64" else 2e-5


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    return oracle


@pytest.fixture(scope="session")
def O(oracle_mod):
    return oracle_mod.restatement()
