"""The tile-centred Laplace path (sctl_amd/csrc/centered_kernel.hpp): Morton-sorted targets, far sources through the
expanded distance, near sources exact.  Checked against the CPU oracle on a target subset and against the exact all-pairs
kernel (SCTL_AMD_CENTERED=0) on ALL targets, for point distributions that stress the far/near split."""
import os

import numpy as np
import pytest

import sctl_amd
from conftest import rel_l2

pytestmark = pytest.mark.gpu

NT, NS = (1 << 18) + 37, 70001      # ragged: not multiples of the 128-target wave or the 64-source tile


def _clouds(kind, rng):
    if kind == "uniform":
        return rng.random((NT, 3)), rng.random((NS, 3))
    if kind == "clustered":          # targets in tight clusters, sources inside and far outside them
        centres = rng.random((50, 3))
        xt = centres[rng.integers(0, 50, NT)] + 1e-3 * rng.standard_normal((NT, 3))
        xs = np.concatenate([centres[rng.integers(0, 50, NS // 2)] + 1e-3 * rng.standard_normal((NS // 2, 3)), 10 * rng.random((NS - NS // 2, 3))])
        return xt, xs
    if kind == "surface":            # both on a sphere (a boundary-integral discretisation), sources = a subset of the targets
        v = rng.standard_normal((NT, 3))
        xt = v / np.linalg.norm(v, axis=1, keepdims=True)
        return xt, xt[rng.choice(NT, NS, replace=False)].copy()
    if kind == "identical_targets":  # zero cluster radius: every source is "far" unless it coincides with the target
        xt = np.tile(np.array([[0.3, 0.4, 0.5]]), (NT, 1))
        xs = rng.random((NS, 3))
        xs[:10] = xt[0]
        return xt, xs
    if kind == "offset":             # large common offset: centring must not lose the small separations
        return 1.0e4 + 1e-2 * rng.random((NT, 3)), 1.0e4 + 1e-2 * rng.random((NS, 3))
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "surface", "identical_targets", "offset"])
@pytest.mark.parametrize("name", ["Laplace3D-FxU", "Laplace3D-DxU", "Laplace3D-FxdU", "Stokes3D-FxUP"])
def test_centred_path_matches_oracle_and_exact_kernel(O, name, kind):
    """(the gradient kernel and the Stokes velocity + pressure kernel, round 4: their far sources are summed as moments, x_t' sum A - sum A x_s'; fp64 only)"""
    import torch
    rng = np.random.default_rng(123)
    xt, xs = _clouds(kind, rng)
    xt, xs = np.ascontiguousarray(xt.ravel()), np.ascontiguousarray(xs.ravel())
    info = sctl_amd.kernel_info(name)
    f = rng.random(NS * info["k0"]) - 0.5
    xn = (rng.random(NS * 3) - 0.5) if info["nd"] else None
    assert sctl_amd.plan(name, 0, NT, NS)["path"] == "tile-centred"
    # (vector outputs in fp32: on the matrix cores, test_fp32_vector_kernels_on_the_matrix_cores)
    assert sctl_amd.plan(name, 1, NT, NS)["path"] == "tile-centred"
    d = [None if a is None else torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d).cpu().numpy()
    assert np.all(np.isfinite(u))
    os.environ["SCTL_AMD_CENTERED"] = "0"
    try:
        assert sctl_amd.plan(name, 0, NT, NS)["path"] == "exact"
        u_exact = sctl_amd.eval_device(name, *d).cpu().numpy()
    finally:
        del os.environ["SCTL_AMD_CENTERED"]
    # the double layer sums signed terms of size 1/r^2: where they nearly cancel the two summation orders differ more
    assert rel_l2(u, u_exact) <= (2e-14 if name.endswith("FxU") else 2e-13), rel_l2(u, u_exact)
    sel = rng.choice(NT, 300, replace=False)
    ref = O.eval(name, xt.reshape(NT, 3)[sel].ravel().copy(), xs, xn, f)
    assert rel_l2(u.reshape(NT, -1)[sel].ravel(), ref) <= 1e-12, rel_l2(u.reshape(NT, -1)[sel].ravel(), ref)
    if info["k1"] > 1:       # accuracy modes and accumulate semantics of the new policies (the scalar ones: the next test)
        v0 = torch.from_numpy(rng.random(NT * info["k1"]) - 0.5).cuda()
        u10 = sctl_amd.eval_device(name, *d, v_trg=v0.clone(), digits=10).cpu().numpy() - v0.cpu().numpy()
        assert rel_l2(u10, u) <= 1e-9 and sctl_amd.plan(name, 0, NT, NS, digits=10)["path"] == "tile-centred"
        u3 = sctl_amd.eval_device(name, *d, digits=3).cpu().numpy()
        assert rel_l2(u3, u) <= 1e-2


def test_centred_path_accumulates_and_honours_digits(O):
    import torch
    rng = np.random.default_rng(5)
    xt, xs, f = rng.random(NT * 3), rng.random(NS * 3), rng.random(NS) - 0.5
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, f)]
    u = sctl_amd.eval_device("Laplace3D-FxU", d[0], d[1], None, d[2])
    u2 = sctl_amd.eval_device("Laplace3D-FxU", d[0], d[1], None, d[2], v_trg=u.clone())          # accumulate (generic-kernel.txx:184)
    assert rel_l2(u2.cpu().numpy(), 2 * u.cpu().numpy()) < 1e-15
    sel = rng.choice(NT, 200, replace=False)
    ref = O.eval("Laplace3D-FxU", xt.reshape(NT, 3)[sel].ravel().copy(), xs, None, f)
    for digits, tol in ((3, 1e-2), (7, 1e-6), (12, 1e-11), (-1, 1e-13)):
        v = sctl_amd.eval_device("Laplace3D-FxU", d[0], d[1], None, d[2], digits=digits).cpu().numpy()
        assert rel_l2(v[sel], ref) <= tol, (digits, rel_l2(v[sel], ref))


@pytest.mark.parametrize("name", ["Laplace3D-FxU", "Laplace3D-DxU"])
def test_centred_path_fp32(O, name):
    """fp32 through the tile-centred path against the fp64 oracle (tolerance of SURVEY.md §8d) and the exact fp32 kernel."""
    import torch
    rng = np.random.default_rng(9)
    n = 1 << 18
    xt, xs, f = rng.random(n * 3).astype(np.float32), rng.random(n * 3).astype(np.float32), (rng.random(n) - 0.5).astype(np.float32)
    xn = (rng.random(n * 3) - 0.5).astype(np.float32) if name.endswith("DxU") else None
    assert sctl_amd.plan(name, 1, n, n)["path"] == "tile-centred"
    d = [None if a is None else torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d).cpu().numpy()
    os.environ["SCTL_AMD_CENTERED"] = "0"
    try:
        u_exact = sctl_amd.eval_device(name, *d).cpu().numpy()
    finally:
        del os.environ["SCTL_AMD_CENTERED"]
    sel = rng.choice(n, 300, replace=False)
    ref = O.eval(name, xt.reshape(n, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), None if xn is None else xn.astype(np.float64), f.astype(np.float64))
    e_c, e_x = rel_l2(u[sel], ref), rel_l2(u_exact[sel], ref)
    assert e_c <= 1e-4 and e_c <= 3 * e_x + 1e-6, (e_c, e_x)


def test_compact_slab_uses_the_centred_path_and_matches(O):
    """A rank's share of a multi-GPU job (sctl_amd_eval_device_slab): 2^17 targets cut from the Morton order of 2^20, all
    the sources.  Same potential as the exact kernel and the oracle; the slab hint only changes the path taken; and the
    one-process-per-GPU driver (world size 1 here) returns the caller's order."""
    import torch
    from sctl_amd.distributed import morton_order
    rng = np.random.default_rng(77)
    n_whole, G, ns = 1 << 20, 8, 1 << 17
    xt_all = torch.from_numpy(rng.random(n_whole * 3)).cuda()
    xs, f = torch.from_numpy(rng.random(ns * 3)).cuda(), torch.from_numpy(rng.random(ns) - 0.5).cuda()
    perm = morton_order(xt_all)
    n = n_whole // G
    assert sctl_amd.plan("Laplace3D-FxU", 0, n, ns)["path"] == "exact"
    assert sctl_amd.plan("Laplace3D-FxU", 0, n, ns, nt_whole=n_whole)["path"] == "tile-centred"
    for g in (0, 3, 7):
        slab = xt_all.view(-1, 3)[perm[g * n:(g + 1) * n]].contiguous().view(-1)
        u = sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, nt_whole=n_whole).cpu().numpy()
        u_exact = sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f).cpu().numpy()
        assert rel_l2(u, u_exact) <= 2e-14, (g, rel_l2(u, u_exact))
        sel = rng.choice(n, 200, replace=False)
        ref = O.eval("Laplace3D-FxU", slab.cpu().numpy().reshape(n, 3)[sel].ravel().copy(), xs.cpu().numpy(), None, f.cpu().numpy())
        assert rel_l2(u[sel], ref) <= 1e-12
    # a "slab" that is NOT in curve order (the hint's contract broken): still the right potential — every wave's cluster is then the whole
    # slab, its sources are all "near" and take the exact path
    shuffle = torch.from_numpy(rng.permutation(n)).cuda()
    u_bad = sctl_amd.eval_device("Laplace3D-FxU", slab.view(-1, 3)[shuffle].contiguous().view(-1), xs[:3 * 65536].contiguous(), None, f[:65536].contiguous(), nt_whole=n_whole)
    u_ok = sctl_amd.eval_device("Laplace3D-FxU", slab, xs[:3 * 65536].contiguous(), None, f[:65536].contiguous())
    assert rel_l2(u_bad.cpu().numpy(), u_ok[shuffle].cpu().numpy()) <= 2e-14
    with pytest.raises(sctl_amd.api.SctlAmdError):
        sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, nt_whole=n - 1)         # a slab larger than its whole


def test_one_process_multi_device_slabs_follow_the_morton_curve(O):
    """sctl_amd_eval_host_multi / sctl_amd_op_* with several devices (the same GPU listed twice on a one-GPU box): slabs are
    cut from the Morton order, evaluated with the slab hint (tile-centred path at 2^17 targets per device out of 2^18), and
    the potential comes back in the caller's order; accumulate and overwrite semantics hold; fp32 too."""
    rng = np.random.default_rng(41)
    n, ns = 1 << 18, 70001
    xt, xs, f = rng.random(n * 3), rng.random(ns * 3), rng.random(ns) - 0.5
    one = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f)
    two = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f, devices=[0, 0])
    assert rel_l2(two, one) <= 2e-14, rel_l2(two, one)
    sel = rng.choice(n, 200, replace=False)
    ref = O.eval("Laplace3D-FxU", xt.reshape(n, 3)[sel].ravel().copy(), xs, None, f)
    assert rel_l2(two[sel], ref) <= 1e-12
    for devs in ((0, 0, 0), (0,)):                        # one device: the handle still keeps the targets Morton-sorted (no per-call sort)
        op = sctl_amd.DirectOp("Laplace3D-FxU", np.float64, devices=devs)
        op.set_targets(xt)
        op.set_sources(xs)
        u = np.full(n, 0.5)
        op.eval(f, u, accumulate=True)
        assert rel_l2(u - 0.5, one) <= 1e-13
        op.eval(f, u, accumulate=False)                  # EvalDirect: overwrite
        assert rel_l2(u, one) <= 2e-14
        op.eval(f, u, accumulate=False, digits=10)       # the unnormalised Newton step on the pre-sorted centred path
        assert rel_l2(u, one) <= 1e-13
    three32 = sctl_amd.eval_host("Laplace3D-FxU", xt.astype(np.float32), xs.astype(np.float32), None, f.astype(np.float32), devices=[0, 0, 0])
    assert rel_l2(three32[sel].astype(np.float64), ref) <= 1e-4


@pytest.mark.parametrize("name,dt", [("Laplace3D-FxdU", np.float64), ("Stokes3D-FxUP", np.float64), ("Stokes3D-FxU", np.float32), ("Stokes3D-FxUP", np.float32),
                                     ("Stokes3D-DxU", np.float32), ("Stokes3D-FxT", np.float32), ("Laplace3D-FxdU", np.float32), ("Laplace3D-FDxUdU", np.float32)])
def test_vector_kernels_through_the_operator_handle_and_slabs(O, name, dt):
    """Kernels with several outputs per target on the tile-centred path (fp64: far moments on the vector pipe; fp32 Stokeslet family: the matrix-core kernel)
    through the entries that keep the targets in Morton order: the operator handle on one device (no per-call sort: the kernel writes K1 values per target in
    sorted order, the handle puts them back) and three slabs of one device list; accumulate and overwrite; against the host-buffer entry and the oracle."""
    rng = np.random.default_rng(43)
    n, ns = (1 << 18) + 5, 70001
    info = sctl_amd.kernel_info(name)
    k1 = info["k1"]
    xt, xs, f = rng.random(n * 3).astype(dt), rng.random(ns * 3).astype(dt), (rng.random(ns * info["k0"]) - 0.5).astype(dt)
    xn = (rng.random(ns * 3) - 0.5).astype(dt) if info["nd"] else None
    f64 = dt == np.float64
    assert sctl_amd.plan(name, 0 if f64 else 1, n, ns)["path"] == "tile-centred"
    one = sctl_amd.eval_host(name, xt, xs, xn, f)
    sel = rng.choice(n, 200, replace=False)
    ref = O.eval(name, xt.reshape(n, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), None if xn is None else xn.astype(np.float64), f.astype(np.float64))
    assert rel_l2(one.reshape(n, k1)[sel].ravel(), ref) <= (1e-12 if f64 else 2e-5)
    for devs in ((0,), (0, 0, 0)):
        op = sctl_amd.DirectOp(name, dt, devices=devs)
        op.set_targets(xt)
        op.set_sources(xs, xn)
        u = np.full(n * k1, 0.25, dtype=dt)
        op.eval(f, u, accumulate=True)
        assert rel_l2(u - dt(0.25), one) <= (1e-13 if f64 else 2e-5), (devs, rel_l2(u - dt(0.25), one))     # (slabs have their own cluster centres: fp32 sums differ in the last bits)
        op.eval(f, u, accumulate=False)
        assert rel_l2(u, one) <= (2e-14 if f64 else 2e-5)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "surface", "identical_targets", "offset"])
@pytest.mark.parametrize("name", ["Laplace3D-FxU", "Laplace3D-DxU"])
def test_centred_fp32_contractions_on_the_matrix_cores_match_the_packed_valu_kernel(O, name, kind):
    """fp32 Laplace single and double layer: the kernel whose far-pair r^2 (and, for the double layer, (x_t - x_s).n f) is a split-bf16 contraction on
    the matrix cores (centered_mfma_kernel.hpp, the default) against the packed-VALU kernel (SCTL_AMD_MFMA_F32=0) and the fp64 oracle on the same
    fp32-rounded inputs, on the point clouds that stress the far / near split; ragged sizes, so the last target tile, the last source tile and the
    carried leftovers are exercised."""
    import torch
    rng = np.random.default_rng(321)
    xt, xs = _clouds(kind, rng)
    xt, xs = np.ascontiguousarray(xt.ravel()).astype(np.float32), np.ascontiguousarray(xs.ravel()).astype(np.float32)
    f = (rng.random(NS) - 0.5).astype(np.float32)
    xn = (rng.random(NS * 3) - 0.5).astype(np.float32) if name.endswith("DxU") else None
    pl = sctl_amd.plan(name, 1, NT, NS)
    assert pl["path"] == "tile-centred" and pl["pipe"].startswith("bf16 matrix cores"), pl
    assert sctl_amd.plan(name, 1, NT, NS, digits=10)["pipe"] == "vector pipe" and sctl_amd.plan(name, 0, NT, NS)["pipe"] == "vector pipe"
    d = [None if a is None else torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d).cpu().numpy()
    assert np.all(np.isfinite(u))
    os.environ["SCTL_AMD_MFMA_F32"] = "0"
    try:
        assert sctl_amd.plan(name, 1, NT, NS)["pipe"] == "vector pipe"
        u_valu = sctl_amd.eval_device(name, *d).cpu().numpy()
    finally:
        del os.environ["SCTL_AMD_MFMA_F32"]
    sel = rng.choice(NT, 300, replace=False)
    ref = O.eval(name, xt.reshape(NT, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), None if xn is None else xn.astype(np.float64),
                 f.astype(np.float64))
    e_m, e_v = rel_l2(u[sel], ref), rel_l2(u_valu[sel], ref)
    # (the double layer's signed 1/r^2 terms nearly cancel on some of these clouds: both kernels then sit further from the fp64 result, together)
    assert e_m <= (1e-4 if name.endswith("FxU") else 1e-3) and e_m <= 2 * e_v + 5e-7, (name, kind, e_m, e_v)   # measured: e_m / e_v <= 1.3 except where both are a few fp32 ulps (7.8e-7 vs 2.9e-7 on the offset cloud)
    assert rel_l2(u, u_valu) <= (2e-5 if name.endswith("FxU") else 2e-4), (name, kind, rel_l2(u, u_valu))


@pytest.mark.parametrize("kind", ["uniform", "clustered", "surface", "identical_targets", "offset"])
@pytest.mark.parametrize("name", ["Stokes3D-FxU", "Stokes3D-FSxU", "Stokes3D-FxUP", "Stokes3D-DxU", "Stokes3D-FxT", "Laplace3D-FxdU", "Laplace3D-FDxUdU"])
def test_fp32_vector_kernels_on_the_matrix_cores(O, name, kind):
    """fp32 Stokeslet, Stokeslet + source/sink, velocity + pressure, the stresslet, the traction tensor, the Laplace gradient (kernel_functions.hpp:53-198) and the fused Laplace kernel at the default accuracy: r^2 AND the dot
    products (x_t - x_s).f, (x_t - x_s).n of the far pairs as split-bf16 contractions on the matrix cores, four far moments per target on the vector pipe
    (centered_mfma_kernel.hpp, round 4) — against the exact fp32
    kernel and the fp64 oracle on the same fp32-rounded inputs, on the clouds that stress the far / near split, ragged sizes; accumulate semantics; more digits than the
    seed's take the exact kernel; six evaluations bit-identical."""
    import torch
    rng = np.random.default_rng(654)
    xt, xs = _clouds(kind, rng)
    xt, xs = np.ascontiguousarray(xt.ravel()).astype(np.float32), np.ascontiguousarray(xs.ravel()).astype(np.float32)
    info = sctl_amd.kernel_info(name)
    f = (rng.random(NS * info["k0"]) - 0.5).astype(np.float32)
    xn = (rng.random(NS * 3) - 0.5).astype(np.float32) if info["nd"] else None
    pl = sctl_amd.plan(name, 1, NT, NS)
    assert pl["path"] == "tile-centred" and pl["pipe"].startswith("bf16 matrix cores") and pl["trg_per_lane"] == 2, pl
    assert sctl_amd.plan(name, 1, NT, NS, digits=9)["path"] == "exact"                      # fp32 beyond the seed's accuracy: the exact kernel's Newton step
    assert sctl_amd.plan(name, 0, NT, NS)["pipe"] == "vector pipe"
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, f)]
    dn = None if xn is None else torch.from_numpy(xn).cuda()
    runs = [sctl_amd.eval_device(name, d[0], d[1], dn, d[2]).clone() for _ in range(6)]
    for r in runs[1:]:
        assert int((r.view(torch.int32) != runs[0].view(torch.int32)).sum()) == 0, (name, kind)
    u = runs[0].cpu().numpy()
    assert np.all(np.isfinite(u))
    v0 = torch.from_numpy((rng.random(NT * info["k1"]) - 0.5).astype(np.float32)).cuda()
    u_acc = sctl_amd.eval_device(name, d[0], d[1], dn, d[2], v_trg=v0.clone()).cpu().numpy()          # v_trg += (generic-kernel.txx:184)
    assert np.max(np.abs((u_acc - v0.cpu().numpy()) - u)) <= 1e-5 * np.max(np.abs(u))
    for env in ("SCTL_AMD_MFMA_F32", "SCTL_AMD_CENTERED"):                                              # either switch leaves the exact fp32 kernel
        os.environ[env] = "0"
        try:
            assert sctl_amd.plan(name, 1, NT, NS)["path"] == "exact"
            u_exact = sctl_amd.eval_device(name, d[0], d[1], dn, d[2]).cpu().numpy()
        finally:
            del os.environ[env]
    sel = rng.choice(NT, 300, replace=False)
    ref = O.eval(name, xt.reshape(NT, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), None if xn is None else xn.astype(np.float64), f.astype(np.float64))
    e_m, e_x = rel_l2(u.reshape(NT, -1)[sel].ravel(), ref), rel_l2(u_exact.reshape(NT, -1)[sel].ravel(), ref)
    # the split-bf16 r^2 (good to ~3e-7 after the far condition's cancellation) enters the 1/r^3 terms three times: 1.5e-6 .. 4e-6 where the exact kernel has 5e-7
    assert e_m <= 2e-5 and e_m <= 10 * e_x + 2e-6, (name, kind, e_m, e_x)
    assert rel_l2(u, u_exact) <= 2e-5, (name, kind, rel_l2(u, u_exact))


@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_centred_kernels_give_bit_identical_results_run_to_run(kind):
    """Every sum runs in a fixed order, so repeated evaluations of one problem must agree to the last bit.  Round 3 found near-field sums of one
    instantiation of the fp32 double-layer kernels that did not: whole contributions lost in lanes 48-63 when packed-fp32 and transcendental instructions
    of the near flush met another wave's transcendental bursts (profiles/r04_near_fault_report.md) — a fault a tolerance test against the
    oracle sees only when it is large, and one that shows only when OTHER kernels run between the launches: here the copy of every result and the fills of
    the next output do that.  EVERY instantiation the library can launch is run: both fp32 pipes, the matrix-core kernels with 256 and with 128 targets per
    wave (SCTL_AMD_MFMA_CB=4 selects the latter, the form that faulted), fp64, single and double layer; the clouds make waves flush their near list between
    tiles (the clustered one in every wave).  tools/kernel_repeat_ranges.py on the frozen reproducer is the long form; tools/check_isa_rules.py (a CPU test)
    holds the compiled code to what the analysis found."""
    import torch
    rng = np.random.default_rng(99)
    xt, xs = _clouds(kind, rng)
    xn = rng.random(NS * 3) - 0.5
    f = rng.random(NS) - 0.5
    for dt in (np.float32, np.float64):
        d = [torch.from_numpy(np.ascontiguousarray(a.ravel()).astype(dt)).cuda() for a in (xt, xs, xn, f)]
        bits = torch.int32 if dt == np.float32 else torch.int64
        # (pipe switch, targets-per-wave switch, targets per lane the plan must report)
        cases = (("1", None, 4), ("1", "8", 4), ("1", "4", 2), ("0", None, 2)) if dt == np.float32 else (("1", None, 4),)
        for mfma, cb, per_lane in cases:
            os.environ["SCTL_AMD_MFMA_F32"] = mfma
            if cb:
                os.environ["SCTL_AMD_MFMA_CB"] = cb
            try:
                for name in ("Laplace3D-FxU", "Laplace3D-DxU") + (("Laplace3D-FxdU", "Stokes3D-FxUP") if dt == np.float64 else ()):
                    pl = sctl_amd.plan(name, 1 if dt == np.float32 else 0, NT, NS)
                    want = 3 if name.endswith("FxdU") else per_lane
                    if dt == np.float32 and mfma == "1" and cb is None and name.endswith("FxU"):
                        want = 2          # the single layer on a target set of under 2^20 points: the 128-target form (fewer near sources)
                        assert sctl_amd.plan(name, 1, 1 << 20, NS)["trg_per_lane"] == 4 and sctl_amd.plan(name, 1, NT, NS, nt_whole=1 << 21)["trg_per_lane"] == 4
                    assert pl["path"] == "tile-centred" and pl["trg_per_lane"] == want, (pl, mfma, cb)
                    assert pl["pipe"].startswith("bf16 matrix cores") == (dt == np.float32 and mfma == "1"), pl
                    dens = d[3] if sctl_amd.kernel_info(name)["k0"] == 1 else d[2]       # (three density components: the array drawn for the normals)
                    runs = [sctl_amd.eval_device(name, d[0], d[1], d[2] if name.endswith("DxU") else None, dens).clone() for _ in range(6)]
                    assert bool(torch.isfinite(runs[0]).all())
                    for r in runs[1:]:
                        assert int((r.view(bits) != runs[0].view(bits)).sum()) == 0, (name, dt.__name__, mfma, cb, kind)
            finally:
                del os.environ["SCTL_AMD_MFMA_F32"]
                os.environ.pop("SCTL_AMD_MFMA_CB", None)


_CENTRED_FORMS = [("Laplace3D-FxU", np.float64), ("Laplace3D-FxU", np.float32), ("Laplace3D-DxU", np.float64), ("Laplace3D-DxU", np.float32), ("Laplace3D-FxdU", np.float64),
                  ("Stokes3D-FxUP", np.float64), ("Stokes3D-FxU", np.float32), ("Stokes3D-FSxU", np.float32), ("Stokes3D-FxUP", np.float32), ("Stokes3D-DxU", np.float32),
                  ("Stokes3D-FxT", np.float32), ("Laplace3D-FxdU", np.float32), ("Laplace3D-FDxUdU", np.float32)]


@pytest.mark.parametrize("seed", range(6))
def test_forced_tile_centred_paths_on_small_ragged_problems(O, seed):
    """SCTL_AMD_CENTERED=1 sends every problem of at least 128 targets and 64 sources down the tile-centred path of its kernel (vector pipe or matrix cores): sizes around
    the 64-source tile, the 32-row contraction block and the 128 / 192 / 256 targets of a wave, a single partly filled wave, coincident points (targets copied from
    sources), scaled clouds, every accuracy the form serves, through device tensors, host buffers and the operator handle — each against the CPU oracle on the same inputs."""
    import torch
    rng = np.random.default_rng(4200 + seed)
    os.environ["SCTL_AMD_CENTERED"] = "1"
    try:
        for _ in range(14):
            name, dt = _CENTRED_FORMS[int(rng.integers(0, len(_CENTRED_FORMS)))]
            info = sctl_amd.kernel_info(name)
            f64 = dt == np.float64
            Nt = int(rng.choice([128, 129, 191, 192, 193, 255, 256, 257, 383, 385, 513])) if rng.random() < 0.5 else int(rng.integers(128, 3000))
            Ns = int(rng.choice([64, 65, 95, 96, 97, 127, 128, 129, 191, 193])) if rng.random() < 0.5 else int(rng.integers(64, 3000))
            digits = int(rng.choice([-1, -1, 12, 9, 5])) if f64 else int(rng.choice([-1, -1, 5]))
            scale = float(rng.choice([1.0, 1e-3, 1e3]))
            xs = (scale * rng.random(Ns * 3)).astype(dt)
            xt = (scale * (rng.random(Nt * 3) if rng.random() < 0.7 else 0.2 * rng.random(Nt * 3) + 0.4)).astype(dt)
            if rng.random() < 0.4:                  # coincident points: every such pair contributes exactly 0 (kernel_functions.hpp:28)
                k = min(Nt, Ns, int(rng.integers(1, 200)))
                xt[:k * 3] = xs[:k * 3]
            xn = (rng.random(Ns * info["nd"]) - 0.5).astype(dt)
            f = (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
            pl = sctl_amd.plan(name, 0 if f64 else 1, Nt, Ns, digits=digits)
            assert pl["path"] == "tile-centred", (name, dt.__name__, Nt, Ns, digits, pl)
            ref = O.eval(name, xt.astype(np.float64), xs.astype(np.float64), xn.astype(np.float64), f.astype(np.float64))
            how = int(rng.integers(0, 3))
            if how == 0:
                u = sctl_amd.eval_host(name, xt, xs, xn, f, digits=digits)
            elif how == 1:
                d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
                u = sctl_amd.eval_device(name, *d, digits=digits).cpu().numpy()
            else:
                op = sctl_amd.DirectOp(name, dt)
                op.set_targets(xt)
                op.set_sources(xs, xn)
                u = op.eval(f, digits=digits)
                op.close()
            tol = (1e-12 if digits < 0 else 10.0 * 10.0 ** (-digits)) if f64 else (3e-5 if digits != 5 else 1e-4)
            err = rel_l2(u, ref)
            assert np.all(np.isfinite(u)) and err <= tol, (name, dt.__name__, Nt, Ns, digits, how, scale, err)
    finally:
        del os.environ["SCTL_AMD_CENTERED"]

