"""BASELINE.json's five configs at their own sizes.  Config 1 (16384^2) in full against the reference's stored output; the
others (the oracle cannot finish N^2 there) on a target subset with ALL sources against reference-made fixtures and the
restatement, plus size-independent properties: linearity in the density, translation invariance, accumulate == sum."""
import numpy as np
import pytest

import sctl_amd
from conftest import ctx_for, fullsize_array, fullsize_inputs, load_fullsize_manifest, rel_l2

pytestmark = pytest.mark.gpu


def _cloud(seed, Nt, Ns, info, dt=np.float64):
    rng = np.random.default_rng(seed)
    return (rng.random(Nt * 3).astype(dt), rng.random(Ns * 3).astype(dt), (rng.random(Ns * info["nd"]) - 0.5).astype(dt),
            (rng.random(Ns * info["k0"]) - 0.5).astype(dt))


FULLSIZE = load_fullsize_manifest()["cases"]
SUBSET_GROUPS = sorted({(c["seed"], c["kernel"]) for c in FULLSIZE if c["kind"].startswith("eval_subset")})


@pytest.mark.parametrize("digits", [-1, 10])
def test_config1_all_targets_against_the_reference(O, digits):
    """BASELINE config 1 — Laplace3D single layer, 16384 x 16384, fp64, the size the reference's CPU path is quoted on (a
    src/test-fmm.cpp:6 / fmm-wrapper.txx:35-92 style driver) — through the three entries a caller can take: host buffers
    (GenericKernel::Eval's replacement), device buffers, and the operator handle ParticleFMM::EvalDirect sits on; every one of
    the 16384 targets against the REAL reference's stored output (GenericKernel::Eval, and ParticleFMM::EvalDirect at
    accuracy 10) and against the restatement.  Bar: rel-L2 1e-12 at full precision, 10 * 10^-10 at digits = 10."""
    import torch
    name = "Laplace3D-FxU"
    case = next(c for c in FULLSIZE if c["kind"] == "eval_all" and c["digits"] == digits)
    info = sctl_amd.kernel_info(name)
    xt, xs, xn, f, _ = fullsize_inputs(case, info)
    N = case["N"]
    assert N == 1 << 14 and xt.size == 3 * N
    gold = fullsize_array(case["key"])
    tol = 1e-12 if digits < 0 else 1e-9
    results = {"eval_host": sctl_amd.eval_host(name, xt, xs, None, f, digits=digits),
               "eval_device": sctl_amd.eval_device(name, torch.from_numpy(xt).cuda(), torch.from_numpy(xs).cuda(), None, torch.from_numpy(f).cuda(),
                                                   digits=digits).cpu().numpy()}
    op = sctl_amd.DirectOp(name, np.float64)
    op.set_targets(xt)
    op.set_sources(xs)
    results["DirectOp"] = op.eval(f, digits=digits)
    op.close()
    ref = O.eval(name, xt, xs, None, f)
    for how, u in results.items():
        assert u.shape == gold.shape and np.all(np.isfinite(u)), how
        assert rel_l2(u, gold) <= tol, (how, rel_l2(u, gold))
        assert rel_l2(u, ref) <= tol, (how, rel_l2(u, ref))
        if digits == 10:
            assert rel_l2(u, fullsize_array("cfg1_laplace_sl_16k_fmm10")) <= tol, how
    # the plan this size takes: the exact kernel with the source range split and the fixed-order reduction
    plan = sctl_amd.plan(name, 0, N, N, digits)
    assert plan["path"] == "exact" and plan["src_splits"] > 1 and plan["trg_per_lane"] == 1, plan


@pytest.mark.parametrize("seed,name", SUBSET_GROUPS, ids=lambda v: str(v))
def test_full_size_target_subset_against_the_reference(O, seed, name):
    """BASELINE configs 2-5 (+ Laplace SL / DL at 2^20) at their FULL sizes, inputs from drand48: the device result at a fixed
    512-target subset against (a) the REAL reference's output for those targets and all sources, stored by
    oracle/gen_golden_fullsize.py, and (b) the restatement.  f64: rel-L2 1e-12 (digits 10: 1e-9).  fp32 (config 4, 2^23 x 2^23,
    ~10 s on one GPU): 1e-4 against the reference's fp64 result on the same fp32-rounded inputs and against its own fp32
    result (which itself is 5.5e-5 from fp64)."""
    import torch
    info = sctl_amd.kernel_info(name)
    cases = [c for c in FULLSIZE if c["seed"] == seed]
    xt, xs, xn, f, sel = fullsize_inputs(cases[0], info)
    N = cases[0]["N"]
    f32 = xt.dtype == np.float32
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    ref = O.eval(name, *[a.astype(np.float64) for a in (xt.reshape(N, 3)[sel].ravel(), xs, xn, f)], ctx=ctx_for(name))
    by_digits = {}          # accuracy request -> (device result at the subset, tolerance): every fixture meets the result of ITS OWN accuracy
    for c in cases:
        if c["digits"] not in by_digits:
            u = sctl_amd.eval_device(name, *d, ctx=ctx_for(name), digits=c["digits"]).cpu().numpy().reshape(N, info["k1"])
            assert np.all(np.isfinite(u))
            tol = 1e-4 if f32 else (1e-9 if c["digits"] == 10 else 1e-12)
            by_digits[c["digits"]] = (u[sel].ravel(), tol)
            assert rel_l2(by_digits[c["digits"]][0], ref) <= tol, (c["key"], "oracle", rel_l2(by_digits[c["digits"]][0], ref))
        us, tol = by_digits[c["digits"]]
        gold = fullsize_array(c["key"])
        assert gold.shape == us.shape
        assert rel_l2(us, gold) <= tol, (c["key"], "reference", rel_l2(us, gold))


@pytest.mark.parametrize("name,Nt,Ns", [("Stokes3D-FxU", 150001, 131075), ("Helmholtz3D-FxU", 70001, 300007), ("Laplace3D-FDxUdU", 33000, 600011),
                                        ("Stokes3D-FSxU", 262147, 65601)])
def test_ragged_sizes_under_the_l2_split_rule(O, name, Nt, Ns):
    """From 2^34 pairs on the planner cuts the sources into splits of <= 2 MB, in eights, each owned by one XCD (the kernel remaps its launch
    index to (tile, split)): ragged target and source counts — a last tile and a last split that are partly empty — against the oracle on a
    target subset that includes the first and the last targets, and accumulate semantics through the partial-sum reduction."""
    import torch
    info = sctl_amd.kernel_info(name)
    p = sctl_amd.plan(name, 0, Nt, Ns)
    assert p["path"] == "exact" and p["src_splits"] >= 8, p          # (a multiple of 8 wherever the tile count allows: the XCD-aware mapping; else the plain one)
    xt, xs, xn, f = _cloud(41, Nt, Ns, info)
    rng = np.random.default_rng(42)
    v0 = rng.random(Nt * info["k1"]) - 0.5
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d, ctx=ctx_for(name), v_trg=torch.from_numpy(v0).cuda()).cpu().numpy().reshape(Nt, info["k1"])
    sel = np.unique(np.concatenate([np.arange(300), np.arange(Nt - 300, Nt), rng.choice(Nt, 400, replace=False)]))
    ref = O.eval(name, xt.reshape(Nt, 3)[sel].ravel().copy(), xs, xn, f, ctx=ctx_for(name)).reshape(-1, info["k1"])
    got = u[sel] - v0.reshape(Nt, info["k1"])[sel]
    assert rel_l2(got, ref) <= 1e-12, rel_l2(got, ref)


def test_linearity_and_translation_invariance_at_scale():
    import torch
    name = "Stokes3D-FxU"
    info = sctl_amd.kernel_info(name)
    N = 1 << 17
    xt, xs, xn, f1 = _cloud(31, N, N, info)
    f2 = np.random.default_rng(32).random(f1.size) - 0.5
    dxt, dxs, df1, df2 = (torch.from_numpy(a).cuda() for a in (xt, xs, f1, f2))
    u1 = sctl_amd.eval_device(name, dxt, dxs, None, df1)
    u2 = sctl_amd.eval_device(name, dxt, dxs, None, df2)
    u12 = sctl_amd.eval_device(name, dxt, dxs, None, 2.0 * df1 - 3.0 * df2)
    assert rel_l2(u12.cpu().numpy(), (2.0 * u1 - 3.0 * u2).cpu().numpy()) < 1e-12
    # translation by a power of two keeps every difference x_t - x_s exact
    shift = 4.0
    us = sctl_amd.eval_device(name, dxt + shift, dxs + shift, None, df1)
    assert rel_l2(us.cpu().numpy(), u1.cpu().numpy()) < 1e-13
    # accumulating into the result of a first call == evaluating twice
    u_acc = sctl_amd.eval_device(name, dxt, dxs, None, df2, v_trg=u1.clone())
    assert rel_l2(u_acc.cpu().numpy(), (u1 + u2).cpu().numpy()) < 1e-14


@pytest.mark.parametrize("N", [1 << 16, 300000])
@pytest.mark.parametrize("name,dt,tol", [("Laplace3D-FxU", np.float64, 1e-12), ("Stokes3D-DxU", np.float64, 1e-12), ("Laplace3D-FxdU", np.float32, 1e-4)])
def test_shuffled_self_interaction_repairs_masked_tiles(O, N, name, dt, tol):
    """targets == sources in SHUFFLED order: every lane meets its coincident source in a different LDS tile.  The kernel
    evaluates tiles without the r = 0 mask, detects the poisoned tile sums and re-runs those tiles masked (N = 300000:
    ~5 % of tiles per wave); at N = 2^16 more than 1/8 of a wave's tiles need repair and it switches to the masked loop."""
    import torch
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(77)
    xs = rng.random(N * 3).astype(dt)
    perm = rng.permutation(N)
    xt = xs.reshape(N, 3)[perm].ravel().copy()
    xn = (rng.random(N * info["nd"]) - 0.5).astype(dt)
    f = (rng.random(N * info["k0"]) - 0.5).astype(dt)
    u = sctl_amd.eval_device(name, *[torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]).cpu().numpy().reshape(N, info["k1"])
    assert np.all(np.isfinite(u))
    sel = rng.choice(N, 256, replace=False)
    ref = O.eval(name, xt.reshape(N, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), xn.astype(np.float64),
                 f.astype(np.float64)).reshape(256, info["k1"])
    assert rel_l2(u[sel], ref) <= tol, rel_l2(u[sel], ref)
