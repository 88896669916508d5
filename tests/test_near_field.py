"""BoundaryIntegralOp near field (reference boundary_integral.txx:816-1012 setup, :1079-1142 apply).
Golden data: tests/golden/near_field.npz, generated from the REAL reference's BoundaryIntegralOp driven with a synthetic
element list (oracle/gen_golden_near.py).  CPU: the oracle's numpy restatement of ComputeNearInterac reproduces the
reference's result from the reference's operator arrays, and far field + near field = ComputePotential.  GPU: the device
routine (sctl_amd_near_*, through the C ABI) does the same."""
import json
import os

import numpy as np
import pytest

import sctl_amd
from conftest import ROOT, rel_l2
from sctl_amd.rand48 import Rand48

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(GOLD, "near_manifest.json")))["cases"]
_NPZ = None


def gold(case, key):
    global _NPZ
    if _NPZ is None:
        _NPZ = np.load(os.path.join(GOLD, "near_field.npz"))
    return _NPZ["%s/%s" % (case["key"], key)]


def near_inputs(c, k0):
    """oracle/gen_golden_near.py:near_inputs — targets, target normals, nodes, node normals, weights, density."""
    g = Rand48(c["seed"])
    xt = g.drand48(c["Nt"] * 3) - 0.5
    xnt = g.drand48(c["Nt"] * 3) - 0.5
    xs = g.drand48(c["Ns"] * 3) - 0.5
    xn = g.drand48(c["Ns"] * 3) - 0.5
    w = g.drand48(c["Ns"]) * 0.01
    f = g.drand48(c["Ns"] * k0) - 0.5
    return xt, xnt, xs, xn, w, f


def dims(O, c):
    inf = O.info(c["kernel"])
    return inf["k0"], (inf["k1"] // 3 if c["trg_normal_dot_prod"] else inf["k1"])


IDS = ["%s-%s" % (c["kernel"], c["key"]) for c in CASES]


_FREE_CACHE = {}


def matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f):
    """Cached per case (several tests and device counts ask for the same one)."""
    if case["key"] not in _FREE_CACHE:
        _FREE_CACHE[case["key"]] = _matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f)
    return _FREE_CACHE[case["key"]].copy()


def _matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f):
    """Contribution of the matrix-free element list of a two-list case (oracle/ref_near_shim.cpp: FreePatchElemList) to the
    potential, in numpy: per element e of that list and near target t,  u[t][k1] += 0.25 sum_{j,k0} f[j][k0] g_j M[(j,k0)][k1],
    g_j = w_j (1 + 0.5 / (1 + |x_t - x_j|^2 / rad^2)), M = scaled kernel matrix of the element's nodes at x_t (dotted with the
    target normal when the operator contracts).  Zero for one-list cases."""
    inf = O.info(case["kernel"])
    k0, k1f = inf["k0"], inf["k1"]
    dot = bool(case["trg_normal_dot_prod"])
    k1 = k1f // 3 if dot else k1f
    ntrg = arrs["near_trg_cnt"].size
    u = np.zeros(ntrg * k1)
    nfree = case.get("free_nodes", 0)
    if not nfree:
        return u
    npe, rad, ns_a = case["nodes_per_elem"], case["rad"], case["Ns"] - nfree
    self_trg = case["Nt"] == 0
    X = xs.reshape(-1, 3) if self_trg else xt.reshape(-1, 3)
    N = (xn if self_trg else xnt).reshape(-1, 3)
    trg_of_entry = np.empty(arrs["near_scatter_index"].size, dtype=np.int64)
    for i in range(ntrg):
        p0, c = int(arrs["near_trg_dsp"][i]), int(arrs["near_trg_cnt"][i])
        trg_of_entry[arrs["near_scatter_index"][p0:p0 + c]] = i
    near_dsp = np.concatenate([[0], np.cumsum(arrs["near_elem_cnt"])])
    first = (ns_a + npe - 1) // npe                        # elements of the matrix list come first (map order: "a_patches" < "b_free")
    for e in range(first, arrs["elem_nds_cnt"].size):
        j0 = ns_a + (e - first) * npe
        j1 = min(j0 + npe, case["Ns"])
        assert j1 - j0 == arrs["elem_nds_cnt"][e] and arrs["K_near_cnt"][e] == 0
        for entry in range(int(near_dsp[e]), int(near_dsp[e + 1])):
            t = trg_of_entry[entry]
            # nthreads=1: hundreds of tiny calls; waking a 128-thread OpenMP team for each costs ~0.1 s on the GPU box's host
            M = O.kernel_matrix(case["kernel"], X[t].copy(), xs[j0 * 3:j1 * 3].copy(), xn[j0 * 3:j1 * 3].copy(), nthreads=1).reshape(j1 - j0, k0, k1f)
            r2 = ((X[t] - xs[j0 * 3:j1 * 3].reshape(-1, 3)) ** 2).sum(1)
            g = w[j0:j1] * (1 + 0.5 / (1 + r2 / rad ** 2))
            if dot:
                M = (M.reshape(j1 - j0, k0, k1, 3) * N[t]).sum(-1)
            u[t * k1:(t + 1) * k1] += 0.25 * np.einsum("j,jk,jkl->l", g, f[j0 * k0:j1 * k0].reshape(-1, k0), M)
    return u


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_near_restatement_matches_reference(O, oracle_mod, case):
    k0, k1 = dims(O, case)
    xt, xnt, xs, xn, w, f = near_inputs(case, k0)
    arrs = {k: gold(case, k) for k in ("elem_nds_cnt", "near_elem_cnt", "K_near_cnt", "K_near", "near_scatter_index", "near_trg_cnt", "near_trg_dsp")}
    assert arrs["near_scatter_index"].size == case["near_entries"] > 0
    un = oracle_mod.near_apply_restatement(k0, k1, F=f, **arrs) + matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f)
    assert rel_l2(un, gold(case, "u_near")) < (1e-15 if not case.get("free_nodes") else 1e-13)
    # ComputePotential = ComputeFarField + ComputeNearInterac (boundary_integral.txx:608-614)
    self_trg = case["Nt"] == 0
    far = oracle_mod.far_field_restatement(O, case["kernel"], None if self_trg else xt, xn if self_trg else xnt, xs, xn, w, f, bool(case["trg_normal_dot_prod"]))
    assert rel_l2(far + un, gold(case, "u_total")) < 1e-10        # the reference's far field ran at tol 1e-10
    # accumulate semantics (:1131-1140)
    u0 = np.full_like(un, 0.25)
    assert rel_l2(oracle_mod.near_apply_restatement(k0, k1, F=f, U=u0.copy(), **arrs) + matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f), un + 0.25) < 1e-14


def test_near_symbols_fail_loudly_without_gpu():
    if sctl_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sctl_amd.api.SctlAmdError, match="no HIP device"):
        sctl_amd.NearOp(1, 1, [1], [1], np.ones(1), [0], [1], [0])


def test_near_create_rejects_inconsistent_arrays():
    bad = [dict(K_near_cnt=[3]),                       # block size != nodes x near targets
           dict(near_trg_cnt=[2]),                     # more entries claimed than the near list holds
           dict(near_scatter_index=[5])]               # permutation out of range
    for kw in bad:
        a = dict(elem_nds_cnt=[2], near_elem_cnt=[1], K_near=np.ones(2), near_scatter_index=[0], near_trg_cnt=[1], near_trg_dsp=[0], K_near_cnt=None)
        a.update(kw)
        with pytest.raises(sctl_amd.api.SctlAmdError) as ei:          # argument checks come before the device is touched
            sctl_amd.NearOp(1, 1, **a)
        assert "no HIP device" not in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_near_device_matches_reference(O, case):
    import torch
    k0, k1 = dims(O, case)
    xt, xnt, xs, xn, w, f = near_inputs(case, k0)
    arrs = {k: gold(case, k) for k in ("elem_nds_cnt", "near_elem_cnt", "K_near_cnt", "K_near", "near_scatter_index", "near_trg_cnt", "near_trg_dsp")}
    op = sctl_amd.NearOp(k0, k1, **arrs)
    assert op.near_entries == case["near_entries"] and op.operator_bytes == arrs["K_near"].size * 8 and op.density_len == f.size
    # the device applies the precomputed matrices; a matrix-free list's share (two-list cases) is the caller's, on the host
    ref = gold(case, "u_near") - matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f)
    u = op.apply(f)
    assert rel_l2(u, ref) < 1e-14, rel_l2(u, ref)
    u2 = op.apply(f, U=u.copy())                                   # accumulates
    assert rel_l2(u2, 2 * ref) < 1e-14
    fd, ud = torch.from_numpy(f).cuda(), torch.full((ref.size,), 0.5, dtype=torch.float64, device="cuda")
    op.apply_device(fd, ud)
    assert rel_l2(ud.cpu().numpy(), ref + 0.5) < 1e-14
    # fp32 operator against the fp64 reference
    arrs32 = dict(arrs, K_near=arrs["K_near"].astype(np.float32))
    u32 = sctl_amd.NearOp(k0, k1, **arrs32).apply(f.astype(np.float32))
    assert rel_l2(u32.astype(np.float64), ref) < 5e-6
    op.close()


@pytest.mark.gpu
def test_near_device_large_random_operator(oracle_mod):
    """Sizes the golden cases do not reach: thousands of elements, blocks wider than one 64-column workgroup and blocks with
    fewer rows than the workgroup has row groups, empty elements, elements without a matrix, targets without near entries."""
    rng = np.random.default_rng(17)
    nelem, ntrg, k0, k1 = 3000, 20000, 3, 3
    nds = rng.integers(0, 9, nelem)
    near = rng.integers(0, 120, nelem)
    near[::97] = 700                                              # a few very wide blocks
    kcnt = nds * near
    kcnt[5::11] = 0                                               # matrix-free elements contribute nothing
    K = rng.standard_normal(int(kcnt.sum()) * k0 * k1)
    n_near = int(near.sum())
    trg_of_entry = rng.integers(0, ntrg // 2, n_near)             # the upper half of the targets has no near entries
    order = np.argsort(trg_of_entry, kind="stable")               # SortScatterIndex: entries grouped by target
    cnt = np.bincount(trg_of_entry, minlength=ntrg)
    dsp = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    F = rng.standard_normal(int(nds.sum()) * k0)
    ref = oracle_mod.near_apply_restatement(k0, k1, nds, near, kcnt, K, order, cnt, dsp, F)
    op = sctl_amd.NearOp(k0, k1, nds, near, K, order, cnt, dsp, K_near_cnt=kcnt)
    u = op.apply(F)
    assert rel_l2(u, ref) < 1e-14, rel_l2(u, ref)
    assert np.all(u.reshape(ntrg, k1)[ntrg // 2:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_boundary_integral_near_field_end_to_end(tmp_path, case):
    """include/sctl_amd/boundary_integral.hpp with a near zone — SetupSelf/SetupNear on the host (single-rank near list,
    user quadrature callbacks, direct part from the device KernelMatrix), ComputeNearInterac on the device — against the REAL
    reference's ComputeNearInterac and ComputePotential for the same synthetic element list (tests/cpp/bie_driver.cpp)."""
    import subprocess
    from test_cpp_host import _build, _read_vector
    exe = _build(tmp_path, "bie_driver")
    out = str(tmp_path / (case["key"] + ".bin"))
    args = [exe, case["kernel"], str(case["seed"]), str(case["Nt"]), str(case["Ns"]), str(case["nodes_per_elem"]), str(case["upsample"]),
            str(case["trg_normal_dot_prod"]), str(int(case["Nt"] == 0)), out, repr(case["rad"]), str(case.get("free_nodes", 0))]
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    u, un = _read_vector(out), _read_vector(out + ".near")
    assert rel_l2(un, gold(case, "u_near")) < 1e-12, rel_l2(un, gold(case, "u_near"))
    assert rel_l2(u, gold(case, "u_total")) < 1e-10, rel_l2(u, gold(case, "u_total"))     # the reference's far field ran at tol 1e-10
    if case["key"] in ("c5", "t1"):      # the same through DeviceSet with three slabs (one GPU listed three times): far + near partitioned by target slab
        p = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, SCTL_AMD_DEVICES="0,0,0"))
        assert p.returncode == 0, p.stderr
        u3 = _read_vector(out)
        assert rel_l2(u3, u) < 1e-14 and rel_l2(u3, gold(case, "u_total")) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("devs", [(0,), (0, 0), (0, 0, 0)], ids=["1dev", "2slabs", "3slabs"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_fused_compute_potential_on_the_operator_handle(O, case, devs):
    """sctl_amd_op_set_near + sctl_amd_op_eval_potential: far field and near field of one BoundaryIntegralOp in one pass over the devices,
    the near operator block-partitioned by target slab (one GPU listed two or three times stands in for a multi-GPU node).  Fed with the
    REFERENCE's near-operator arrays; expected = the reference's ComputePotential (minus a matrix-free list's share, which is host work)."""
    k0, k1 = dims(O, case)
    xt, xnt, xs, xn, w, f = near_inputs(case, k0)
    arrs = {k: gold(case, k) for k in ("elem_nds_cnt", "near_elem_cnt", "K_near_cnt", "K_near", "near_scatter_index", "near_trg_cnt", "near_trg_dsp")}
    self_trg = case["Nt"] == 0
    T, Tn = (xs, xn) if self_trg else (xt, xnt)
    ups = case["upsample"]
    info = O.info(case["kernel"])
    # the far-field quadrature of PatchElemList: every node repeated `ups` times with weight w / ups, density copied to the copies
    x_far, n_far = np.repeat(xs.reshape(-1, 3), ups, 0).ravel(), np.repeat(xn.reshape(-1, 3), ups, 0).ravel()
    w_far, f_far = np.repeat(w / ups, ups), np.repeat(f.reshape(-1, k0), ups, 0).ravel()
    op = sctl_amd.DirectOp(case["kernel"], np.float64, devices=devs)
    op.set_targets(T)
    op.set_sources(x_far, n_far if info["nd"] else None)
    op.set_source_weights(w_far)
    if case["trg_normal_dot_prod"]:
        op.set_target_normals(Tn)
    op.set_near(k1, arrs["elem_nds_cnt"], arrs["near_elem_cnt"], arrs["K_near"], arrs["near_scatter_index"], arrs["near_trg_cnt"], arrs["near_trg_dsp"],
                K_near_cnt=arrs["K_near_cnt"])
    expect = gold(case, "u_total") - matrix_free_part(O, case, arrs, xt, xnt, xs, xn, w, f)
    u = op.eval_potential(f_far, f, digits=11)
    assert rel_l2(u, expect) < 1e-10, rel_l2(u, expect)                    # the reference's far field ran at tol 1e-10
    far = op.eval(f_far, digits=11)                                        # the two legs separately agree to rounding
    near = sctl_amd.NearOp(k0, k1, **arrs).apply(f)
    assert rel_l2(u, far + near) < 1e-14
    u2 = op.eval_potential(f_far, f, v_trg=u.copy(), accumulate=True, digits=11)
    assert rel_l2(u2, 2 * u) < 1e-15
    op.set_targets(T)                                                      # new targets drop the attached operator
    with pytest.raises(sctl_amd.api.SctlAmdError, match="no near-field operator"):
        op.eval_potential(f_far, f)
    op.close()
