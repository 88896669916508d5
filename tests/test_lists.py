"""Batched list evaluation (sctl_amd_lists_*, SURVEY.md §8f row 4, second half): many (target range x source range) direct sums in
one launch — the P2P shape of fmm-wrapper.txx:756-786.  Golden: tests/golden/p2p_lists.npz, the REAL reference's Eval called once
per list (oracle/gen_golden_lists.py).  CPU: the oracle's per-list loop reproduces the reference; argument checks of the plan.
GPU: the HIP path through the C ABI against both."""
import json
import os

import numpy as np
import pytest

import sctl_amd
from conftest import ROOT, rel_l2
from sctl_amd.lists import grid_neighbour_lists, points_in_boxes

GOLD = os.path.join(ROOT, "tests", "golden")
MAN = json.load(open(os.path.join(GOLD, "lists_manifest.json")))
CASES = MAN["cases"]
IDS = ["%s-%s" % (c["kernel"], c["key"]) for c in CASES]
_NPZ = None


@pytest.fixture(params=["packed", "packed up to 64", "one range per wave"])
def small_ranges(request):
    """Target ranges of up to 32 points run PACKED, several per wave, with a flat source index list (the default; SCTL_AMD_LISTS_PACK=64 also packs those of
    33 .. 64 points, two per wave), or — SCTL_AMD_LISTS_PACK=0, and by itself when that list would pass 4 GB or the sources 2^32 — one range per wave as lane
    replicas; every form stays tested."""
    if request.param == "packed":
        os.environ.pop("SCTL_AMD_LISTS_PACK", None)
        yield True
    else:
        os.environ["SCTL_AMD_LISTS_PACK"] = "64" if request.param.endswith("64") else "0"
        try:
            yield request.param.endswith("64")
        finally:
            del os.environ["SCTL_AMD_LISTS_PACK"]


def gold(case):
    global _NPZ
    if _NPZ is None:
        _NPZ = np.load(os.path.join(GOLD, "p2p_lists.npz"))
    return _NPZ[case["key"]]


def case_data(case, info):
    """oracle/gen_golden_lists.py:list_case_inputs — counts, points in their boxes, normals, densities; then the neighbour lists."""
    dt = np.float64 if case["dtype"] == "f64" else np.float32
    rng = np.random.default_rng(case["seed"])
    nb = case["grid"] ** 3
    cs = rng.integers(1, case["max_pts"] + 1, nb)
    ct = cs if case["self_targets"] else rng.integers(1, case["max_pts"] + 1, nb)
    xs = points_in_boxes(case["grid"], cs, rng, dt)
    xt = xs if case["self_targets"] else points_in_boxes(case["grid"], ct, rng, dt)
    ns = int(cs.sum())
    xn = (rng.random(ns * info["nd"]) - 0.5).astype(dt)
    f = (rng.random(ns * info["k0"]) - 0.5).astype(dt)
    lists = grid_neighbour_lists(case["grid"], ct, cs)
    assert lists[0].size == case["nlists"] and int((lists[1] * lists[3]).sum()) == case["pairs"]
    ctx = np.array(MAN["helmholtz_k"]) if case["kernel"].startswith("Helmholtz") else None
    return lists, xt, xs, xn, f, ctx


def oracle_lists(O, name, lists, xt, xs, xn, f, ctx=None, u=None):
    """The oracle = a loop of O.eval over the lists, each accumulating into its target range (f64 arithmetic)."""
    info = O.info(name)
    k0, k1, nd = info["k0"], info["k1"], info["nd"]
    x64 = [a.astype(np.float64) for a in (xt, xs, xn, f)]
    if u is None:
        u = np.zeros(xt.size // 3 * k1)
    for t0, tc, s0, sc in zip(*lists):
        t1, s1 = t0 + tc, s0 + sc
        if tc == 0 or sc == 0:
            continue
        O.eval(name, x64[0][t0 * 3:t1 * 3].copy(), x64[1][s0 * 3:s1 * 3].copy(), x64[2][s0 * nd:s1 * nd].copy(), x64[3][s0 * k0:s1 * k0].copy(),
               v_trg=u[t0 * k1:t1 * k1], ctx=ctx, nthreads=1)
    return u


def tol(case):
    if case["digits"] >= 0:
        return 10.0 * 10.0 ** (-case["digits"])
    return 1e-12 if case["dtype"] == "f64" else 2e-5


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_list_loop_matches_reference(O, case):
    lists, xt, xs, xn, f, ctx = case_data(case, O.info(case["kernel"]))
    assert rel_l2(oracle_lists(O, case["kernel"], lists, xt, xs, xn, f, ctx), gold(case)) <= tol(case)


def test_neighbour_lists_structure():
    rng = np.random.default_rng(1)
    ct, cs = rng.integers(0, 5, 27), rng.integers(0, 5, 27)
    to, tc, so, sc = grid_neighbour_lists(3, ct, cs)
    assert to.size == 343                                  # sum over the 27 boxes of (neighbours + itself) = 7^3
    centre = (to == np.concatenate([[0], np.cumsum(ct)])[13]) & (tc == ct[13])
    assert ct[13] == 0 or centre.sum() >= 27
    x = points_in_boxes(3, ct, rng).reshape(-1, 3)
    box = np.repeat(np.arange(27), ct)
    assert np.all(np.floor(x[:, 0] * 3) == box // 9) and np.all(np.floor(x[:, 2] * 3) == box % 3)


def test_plan_argument_checks_need_no_device():
    i8 = lambda *v: np.array(v, dtype=np.int64)
    mk = lambda to, tc, so, sc, Nt=10, Ns=10: sctl_amd.ListsPlan("Laplace3D-FxU", np.float64, to, tc, so, sc, Nt, Ns)
    p = mk(i8(0, 5), i8(0, 3), i8(0, 0), i8(4, 0))           # only empty lists: a legal plan without work, no GPU needed
    assert p.pairs == 0 and p.work_items == 0
    assert np.array_equal(p.eval_host(np.zeros(30), np.ones(30), None, np.ones(10)), np.zeros(10))
    for bad in ((i8(0, 2), i8(4, 4), i8(0, 0), i8(1, 1)),    # [0,4) and [2,6): overlapping, not identical
                (i8(0, 0), i8(4, 3), i8(0, 0), i8(1, 1)),    # same start, different length
                (i8(8), i8(4), i8(0), i8(1)),                # targets beyond Nt
                (i8(0), i8(4), i8(9), i8(2)),                # sources beyond Ns
                (i8(0), i8(-1), i8(0), i8(1))):              # negative count
        with pytest.raises(sctl_amd.api.SctlAmdError) as ei:
            mk(*bad)
        assert "no HIP device" not in str(ei.value)
    if sctl_amd.device_count() == 0:                         # real work without a GPU: refused, there is no CPU path
        with pytest.raises(sctl_amd.api.SctlAmdError, match="no HIP device"):
            mk(i8(0), i8(4), i8(0), i8(4))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_lists_match_reference_and_oracle(O, case, small_ranges):
    import torch
    name = case["kernel"]
    info = sctl_amd.kernel_info(name)
    lists, xt, xs, xn, f, ctx = case_data(case, info)
    dt = xt.dtype
    plan = sctl_amd.ListsPlan(name, dt, *lists, case["Nt"], case["Ns"], ctx=ctx)
    assert plan.pairs == case["pairs"] and plan.source_ranges == case["nlists"]
    assert plan.work_items >= (case["grid"] ** 3 // 8 if small_ranges else case["grid"] ** 3)      # packed: up to eight small boxes per wave
    u = plan.eval_host(xt, xs, xn, f, digits=case["digits"])
    assert np.all(np.isfinite(u))
    assert rel_l2(u, gold(case)) <= tol(case), rel_l2(u, gold(case))                       # vs the reference
    ref = oracle_lists(O, name, lists, xt, xs, xn, f, ctx)
    assert rel_l2(u, ref) <= tol(case), rel_l2(u, ref)                                     # vs the oracle, all targets
    u2 = plan.eval_host(xt, xs, xn, f, v_trg=u.copy(), digits=case["digits"])           # a right-sized output is accumulated into
    assert rel_l2(u2, 2 * u) <= (1e-6 if dt == np.float32 else 1e-15)
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    ud = torch.full((case["Nt"] * info["k1"],), 0.5, dtype=d[0].dtype, device="cuda")
    plan.eval_device(*d, v_trg=ud, digits=case["digits"])
    assert rel_l2(ud.cpu().numpy(), u.astype(np.float64) + 0.5) <= (1e-15 if dt == np.float64 else 1e-6)   # device entry accumulates too
    pc0 = sctl_amd.counters()["pair_interactions"]
    one_shot = sctl_amd.eval_lists_host(name, *lists, xt, xs, xn, f, digits=case["digits"], ctx=ctx)
    assert np.array_equal(one_shot, u) and sctl_amd.counters()["pair_interactions"] - pc0 == case["pairs"]
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_hip_lists_random_ragged(O, seed, small_ranges):
    """Not a grid: target ranges of random lengths (1 .. 700, some empty), each with a random number (0 .. 40) of source ranges of random
    lengths (0 .. 300, overlapping freely, many of 1-3 points so that LDS tiles span several ranges), random kernel and precision."""
    rng = np.random.default_rng(4000 + seed)
    name = sctl_amd.KERNEL_NAMES[int(rng.integers(0, len(sctl_amd.KERNEL_NAMES)))]
    info = sctl_amd.kernel_info(name)
    dt = np.float64 if rng.random() < 0.7 else np.float32
    Ns = int(rng.integers(50, 3000))
    nbox = int(rng.integers(1, 60))
    tlen = rng.integers(0, 700, nbox) if seed % 2 else rng.integers(0, 40, nbox)
    gaps = rng.integers(0, 5, nbox)                       # targets that belong to no list stay untouched
    tstart = np.cumsum(gaps + np.concatenate([[0], tlen[:-1]]))
    Nt = int(tstart[-1] + tlen[-1] + 3)
    to, tc, so, sc = [], [], [], []
    for b in range(nbox):
        for _ in range(int(rng.integers(0, 41))):
            n = int(rng.integers(0, 4)) if rng.random() < 0.5 else int(rng.integers(0, min(300, Ns)))
            s0 = int(rng.integers(0, Ns - n + 1))
            to.append(tstart[b]); tc.append(tlen[b]); so.append(s0); sc.append(n)
    order = rng.permutation(len(to))                      # lists arrive in any order; the order INSIDE a target range is kept
    lists = [np.array(a, dtype=np.int64)[order] for a in (to, tc, so, sc)]
    xt, xs = rng.random(Nt * 3).astype(dt), rng.random(Ns * 3).astype(dt)
    xn, f = (rng.random(Ns * info["nd"]) - 0.5).astype(dt), (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
    ctx = np.array([3.0, 0.2]) if name.startswith("Helmholtz") else None
    u0 = rng.random(Nt * info["k1"]).astype(dt)
    u = sctl_amd.eval_lists_host(name, *lists, xt, xs, xn, f, v_trg=u0.copy(), ctx=ctx)
    ref = oracle_lists(O, name, lists, xt, xs, xn, f, ctx, u=u0.astype(np.float64).copy())
    assert rel_l2(u, ref) <= (1e-12 if dt == np.float64 else 3e-5), (name, dt, rel_l2(u, ref))
    covered = np.zeros(Nt, dtype=bool)
    for t0, n, m in zip(lists[0], lists[1], lists[3]):
        if n and m:
            covered[t0:t0 + n] = True
    assert np.array_equal(u.reshape(Nt, -1)[~covered], u0.reshape(Nt, -1)[~covered])      # untouched targets keep their bits


@pytest.mark.gpu
@pytest.mark.parametrize("name", sctl_amd.KERNEL_NAMES)
def test_hip_lists_every_kernel_every_item_shape(O, name, small_ranges):
    """Every kernel through every shape of a work item — two targets per lane (200 targets), and for the small ranges (50, 5, 20, 12, 33, 64, 1 targets) the four
    packed classes or, unpacked, one target per lane and lane replicas — with a box acting on itself; the traction kernel's mirrored output (ukernels.hpp:
    finish) is checked entry by entry."""
    rng = np.random.default_rng(77)
    info = sctl_amd.kernel_info(name)
    tlen = np.array([200, 50, 5, 20, 12, 33, 64, 1], dtype=np.int64)
    tstart = np.concatenate([[0], np.cumsum(tlen)[:-1]])
    Nt, Ns = int(tlen.sum()), 700
    xt = rng.random(Nt * 3)
    xs = np.concatenate([xt, rng.random((Ns - Nt) * 3)])      # the first Nt sources ARE the targets (r = 0 pairs in the "self" lists)
    xn, f = rng.random(Ns * info["nd"]) - 0.5, rng.random(Ns * info["k0"]) - 0.5
    to, tc, so, sc = [], [], [], []
    for b in range(tlen.size):
        for s0, n in ((int(tstart[b]), int(tlen[b])), (300, 173), (473, 64), (650, 3))[:4 if b % 3 else 3]:   # itself, then two or three other ranges
            to.append(tstart[b]); tc.append(tlen[b]); so.append(s0); sc.append(n)
    lists = [np.array(a, dtype=np.int64) for a in (to, tc, so, sc)]
    ctx = np.array([3.0, 0.2]) if name.startswith("Helmholtz") else None
    u = sctl_amd.eval_lists_host(name, *lists, xt, xs, xn, f, ctx=ctx)
    ref = oracle_lists(O, name, lists, xt, xs, xn, f, ctx, u=np.zeros(Nt * info["k1"]))
    assert rel_l2(u, ref) <= 1e-12, (name, rel_l2(u, ref))
    if name == "Stokes3D-FxT":
        m = u.reshape(Nt, 3, 3)
        assert np.array_equal(m, m.transpose(0, 2, 1)) and np.abs(m - ref.reshape(Nt, 3, 3)).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.gpu
def test_hip_lists_equal_plain_eval_for_one_list(O):
    """One list covering everything is GenericKernel::Eval: same numbers as the all-pairs entry (to summation order)."""
    rng = np.random.default_rng(9)
    Nt, Ns = 1000, 3000
    xt, xs, f = rng.random(Nt * 3), rng.random(Ns * 3), rng.random(Ns * 3) - 0.5
    one = [np.array([v], dtype=np.int64) for v in (0, Nt, 0, Ns)]
    a = sctl_amd.eval_lists_host("Stokes3D-FxU", *one, xt, xs, None, f)
    b = sctl_amd.eval_host("Stokes3D-FxU", xt, xs, None, f)
    assert rel_l2(a, b) < 1e-14


@pytest.mark.gpu
def test_one_shot_device_entry(O):
    """sctl_amd_eval_lists_device: device arrays, plan built and released inside the call, returns after the stream has finished."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(3)
    Nt, Ns = 900, 1200
    xt, xs, f = rng.random(Nt * 3), rng.random(Ns * 3), rng.random(Ns) - 0.5
    lists = [np.array(v, dtype=np.int64) for v in ([0, 0, 300, 700], [300, 300, 400, 200], [0, 600, 100, 0], [600, 600, 900, 1200])]
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, f)]
    u = torch.zeros(Nt, dtype=torch.float64, device="cuda")
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = sctl_amd.lib().sctl_amd_eval_lists_device(0, 0, 4, p(lists[0]), p(lists[1]), p(lists[2]), p(lists[3]), Nt, Ns, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()),
                                                   None, C.c_void_p(d[2].data_ptr()), C.c_void_p(u.data_ptr()), -1, None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, sctl_amd.last_error()
    ref = oracle_lists(O, "Laplace3D-FxU", lists, xt, xs, np.zeros(0), f)
    assert rel_l2(u.cpu().numpy(), ref) < 1e-13
