"""INTEGRATION.md's reference-side binding, EXECUTED: the reference's own code — the type-erased GenericKernel::Eval entry,
ParticleFMM<double,3>::EvalDirect (fmm-wrapper.txx:490-562), BoundaryIntegralOp::ComputePotential with and without a near zone
(boundary_integral.txx:608-614; SetupNear's KernelMatrix calls inside its OpenMP loop, :949-986) — compiled in the build container
from the headers where they lie, with ONE change: the kernel class is include/sctl_amd/sctl_dropin.hpp's HipKernel<uKernel>, whose
entries call libsctl_amd.so (oracle/Makefile `ref`: libsctl_ref_dropin.so / libsctl_ref_near_dropin.so, built with -DSCTL_REF_DROPIN).
Expected values: the golden outputs the UNMODIFIED reference produced for the same seeds (tests/golden/)."""
import json
import os

import numpy as np
import pytest

import sctl_amd
from conftest import ROOT, case_inputs, ctx_for, golden_array, load_manifest, rel_l2, tol_for
from test_near_field import CASES as NEAR_CASES, IDS as NEAR_IDS, gold as near_gold, near_inputs

pytestmark = pytest.mark.gpu

MAN = load_manifest()["cases"]


@pytest.fixture(scope="module")
def D(oracle_mod):
    sctl_amd.lib()                      # libsctl_amd.so (and the process's one HIP runtime) first; the shim then binds to it
    d = oracle_mod.dropin()
    if d is None:
        pytest.skip("oracle/_ref/libsctl_ref_dropin.so was not built (no reference tree where build() ran)")
    return d


def _pick(kind, every=1):
    return [c for c in MAN if c["kind"] == kind][::every]


@pytest.mark.parametrize("case", _pick("eval", 3) + _pick("eval_self", 4) + _pick("eval_accumulate", 2), ids=lambda c: "%s-%s" % (c["kernel"], c["key"]))
def test_reference_static_eval_entry_on_the_gpu(D, case):
    """GenericKernel's type-erased static Eval (generic-kernel.hpp:110) as ParticleFMM stores it, fp64 and fp32, incl. targets ==
    sources, pre-filled output (accumulate) and the digits argument."""
    info = D.info(case["kernel"])
    xt, xs, xn, f, v0 = case_inputs(case, info)
    u = D.eval(case["kernel"], xt, xs, xn, f, v_trg=None if v0 is None else v0.copy(), ctx=ctx_for(case["kernel"]), digits=case["digits"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) <= tol_for(case)


@pytest.mark.parametrize("case", _pick("particle_fmm"), ids=lambda c: c["kernel"])
def test_reference_particle_fmm_eval_direct_on_the_gpu(D, case):
    info = D.info(case["kernel"])
    xt, xs, xn, f, _ = case_inputs(case, info)
    u = D.particle_fmm_eval_direct(case["kernel"], xt, xs, xn, f, digits=case["digits"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) <= 1e-9          # both sides ran at 10 digits


@pytest.mark.parametrize("case", _pick("matrix", 2), ids=lambda c: "%s-%s" % (c["kernel"], c["key"]))
def test_reference_kernel_matrix_on_the_gpu(D, case):
    info = D.info(case["kernel"])
    xt, xs, xn, f, _ = case_inputs(case, info)
    M = D.kernel_matrix(case["kernel"], xt, xs, xn, ctx=ctx_for(case["kernel"]))
    assert rel_l2(M, golden_array(case["kernel"], case["key"])) <= tol_for(case)


@pytest.mark.parametrize("case", _pick("far_field"), ids=lambda c: "%s-%s" % (c["kernel"], c["key"]))
def test_reference_boundary_integral_far_field_on_the_gpu(D, case):
    from test_cpp_host import _far_inputs
    xt, xnt, xs, xn, w, f = _far_inputs(case, D.info(case["kernel"])["k0"])
    u = D.boundary_far_field(case["kernel"], None if case["self_targets"] else xt, xnt, xs, xn, w, f, trg_normal_dot_prod=bool(case["trg_normal_dot_prod"]),
                             tol=1e-10, nodes_per_elem=case["nodes_per_elem"], upsample=case["upsample"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) < 1e-10


@pytest.mark.parametrize("case", NEAR_CASES, ids=NEAR_IDS)
def test_reference_boundary_integral_with_near_zone_on_the_gpu(oracle_mod, D, case):
    """The reference's SetupSelf / SetupNear / ComputeNearInterac / ComputePotential with HipKernel: K_near (whose direct part comes from
    sctl_amd_kernel_matrix_host, called from inside the reference's OpenMP loop), the near lists and both potentials."""
    k0 = D.info(case["kernel"])["k0"]
    xt, xnt, xs, xn, w, f = near_inputs(case, k0)
    r = oracle_mod.reference_near(case["kernel"], xt if case["Nt"] else None, xnt if case["Nt"] else None, xs, xn, w, f, bool(case["trg_normal_dot_prod"]),
                                  1e-10, case["nodes_per_elem"], case["upsample"], case["rad"], free_nodes=case.get("free_nodes", 0), dropin=True)
    for k in ("elem_nds_cnt", "near_elem_cnt", "K_near_cnt", "near_scatter_index", "near_trg_cnt", "near_trg_dsp"):
        assert np.array_equal(r[k], near_gold(case, k)), k
    assert rel_l2(r["K_near"], near_gold(case, "K_near")) < 1e-12
    assert rel_l2(r["u_near"], near_gold(case, "u_near")) < 1e-12
    assert rel_l2(r["u_total"], near_gold(case, "u_total")) < 1e-10


# ---- all GPUs of the node from the one calling process (sctl_amd::Devices() of sctl_dropin.hpp) ------------------------------------
@pytest.fixture()
def three_slabs(D):
    """A three-entry device list — on a one-GPU box three target slabs on device 0, on a node its first GPUs — with the
    work-per-GPU threshold switched off so that the small golden cases really take sctl_amd_eval_host_multi."""
    import torch
    n_gpu = torch.cuda.device_count()
    before, min_pairs = D.get_devices()
    devs = [g % n_gpu for g in range(3)]
    assert D.set_devices(devs, 0) == 3 and D.get_devices() == (devs, 0)
    yield devs
    D.set_devices(before, min_pairs)


def test_dropin_default_device_list_is_the_whole_node(D):
    """No SCTL_AMD_DEVICES and no launcher rank variable in this process: every visible GPU, and 2^32 pairs per GPU before a second one is used."""
    import torch
    if any(v in os.environ for v in ("SCTL_AMD_DEVICES", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK")):
        pytest.skip("the environment pins the device list")
    devs, min_pairs = D.get_devices()
    assert devs == list(range(torch.cuda.device_count())) and min_pairs == 1 << 32


@pytest.mark.parametrize("case", _pick("eval", 4) + _pick("eval_self", 5) + _pick("eval_accumulate", 3), ids=lambda c: "%s-%s" % (c["kernel"], c["key"]))
def test_reference_static_eval_entry_over_a_device_list(D, three_slabs, case):
    """The same entry with the targets block-partitioned over a three-entry device list (fmm-wrapper.txx:507's formula; ragged and tiny
    target counts included: Nt = 1 and 7 leave slabs empty), accumulate semantics and the context blob kept."""
    info = D.info(case["kernel"])
    xt, xs, xn, f, v0 = case_inputs(case, info)
    sctl_amd.reset_counters()
    u = D.eval(case["kernel"], xt, xs, xn, f, v_trg=None if v0 is None else v0.copy(), ctx=ctx_for(case["kernel"]), digits=case["digits"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) <= tol_for(case)
    assert sctl_amd.counters()["pair_interactions"] == case["Nt"] * case["Ns"]


@pytest.mark.parametrize("case", _pick("particle_fmm"), ids=lambda c: c["kernel"])
def test_reference_particle_fmm_eval_direct_over_a_device_list(D, three_slabs, case):
    """The UNMODIFIED reference's ParticleFMM::EvalDirect, one process, targets spread over the device list."""
    info = D.info(case["kernel"])
    xt, xs, xn, f, _ = case_inputs(case, info)
    u = D.particle_fmm_eval_direct(case["kernel"], xt, xs, xn, f, digits=case["digits"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) <= 1e-9


@pytest.mark.parametrize("case", _pick("far_field")[::2], ids=lambda c: "%s-%s" % (c["kernel"], c["key"]))
def test_reference_boundary_integral_far_field_over_a_device_list(D, three_slabs, case):
    from test_cpp_host import _far_inputs
    xt, xnt, xs, xn, w, f = _far_inputs(case, D.info(case["kernel"])["k0"])
    u = D.boundary_far_field(case["kernel"], None if case["self_targets"] else xt, xnt, xs, xn, w, f, trg_normal_dot_prod=bool(case["trg_normal_dot_prod"]),
                             tol=1e-10, nodes_per_elem=case["nodes_per_elem"], upsample=case["upsample"])
    assert rel_l2(u, golden_array(case["kernel"], case["key"])) < 1e-10


@pytest.mark.parametrize("case", NEAR_CASES[::3], ids=NEAR_IDS[::3])
def test_reference_boundary_integral_with_near_zone_over_a_device_list(oracle_mod, D, case):
    """ComputePotential with a near zone: the far field over three slabs, SetupNear's KernelMatrix calls dealt over the list by OpenMP thread."""
    import torch
    k0 = D.info(case["kernel"])["k0"]
    xt, xnt, xs, xn, w, f = near_inputs(case, k0)
    devs = [g % torch.cuda.device_count() for g in range(3)]
    r = oracle_mod.reference_near(case["kernel"], xt if case["Nt"] else None, xnt if case["Nt"] else None, xs, xn, w, f, bool(case["trg_normal_dot_prod"]),
                                  1e-10, case["nodes_per_elem"], case["upsample"], case["rad"], free_nodes=case.get("free_nodes", 0), dropin=True,
                                  dropin_devices=devs)
    assert rel_l2(r["K_near"], near_gold(case, "K_near")) < 1e-12
    assert rel_l2(r["u_total"], near_gold(case, "u_total")) < 1e-10


def test_dropin_uses_one_gpu_for_small_work_and_all_for_large(D):
    """The work-per-GPU rule: with the default threshold a 3000 x 3000 evaluation stays on one GPU (one launch plan), a 2^17 x 2^17 one
    (2^34 pairs >= 3 x 2^32) is cut into three slabs; both equal the single-device library result."""
    import torch
    n_gpu = torch.cuda.device_count()
    before, min_pairs = D.get_devices()
    try:
        D.set_devices([g % n_gpu for g in range(3)], 1 << 32)
        rng = np.random.default_rng(5)
        for N in (3000, 1 << 17):
            xt, xs, f = rng.random(N * 3), rng.random(N * 3), rng.random(N) - 0.5
            u = D.eval("Laplace3D-FxU", xt, xs, None, f)
            ref = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f)
            assert rel_l2(u, ref) <= 1e-14
    finally:
        D.set_devices(before, min_pairs)
