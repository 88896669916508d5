"""Parity of the HIP path (through the C ABI of libsctl_amd.so) with the reference's stored outputs and the CPU oracle.
Every test here needs a real MI355X; the driver runs them with `-m gpu`."""
import numpy as np
import pytest

import sctl_amd
from conftest import case_inputs, ctx_for, golden_array, load_manifest, rel_l2, tol_for

pytestmark = pytest.mark.gpu

MANIFEST = load_manifest()
EVAL_CASES = [c for c in MANIFEST["cases"] if c["kind"] in ("eval", "eval_self", "eval_accumulate", "particle_fmm")]
MATRIX_CASES = [c for c in MANIFEST["cases"] if c["kind"] == "matrix"]


def _id(c):
    return "%s-%s" % (c["kernel"], c["key"])


def _rng_inputs(rng, Nt, Ns, info, dt):
    xt = rng.random(Nt * 3).astype(dt)
    xs = rng.random(Ns * 3).astype(dt)
    xn = (rng.random(Ns * info["nd"]) - 0.5).astype(dt)
    f = (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
    return xt, xs, xn, f


@pytest.mark.parametrize("case", EVAL_CASES, ids=_id)
def test_hip_matches_reference_golden(case):
    """Host-buffer C ABI (the GenericKernel::Eval drop-in) against the REAL reference's output on the same inputs."""
    info = sctl_amd.kernel_info(case["kernel"])
    xt, xs, xn, f, v0 = case_inputs(case, info)
    u = sctl_amd.eval_host(case["kernel"], xt, xs, xn, f, v_trg=None if v0 is None else v0.copy(), digits=case["digits"],
                           ctx=ctx_for(case["kernel"]))
    ref = golden_array(case["kernel"], case["key"])
    assert u.shape == ref.shape
    assert np.all(np.isfinite(u))
    assert rel_l2(u, ref) <= tol_for(case), rel_l2(u, ref)


@pytest.mark.parametrize("case", MATRIX_CASES, ids=_id)
def test_hip_kernel_matrix_matches_reference(case):
    info = sctl_amd.kernel_info(case["kernel"])
    xt, xs, xn, f, _ = case_inputs(case, info)
    M = sctl_amd.kernel_matrix_host(case["kernel"], xt, xs, xn, ctx=ctx_for(case["kernel"]))
    ref = golden_array(case["kernel"], case["key"])
    assert M.shape == ref.shape
    assert rel_l2(M, ref) <= 1e-12


@pytest.mark.parametrize("name", sctl_amd.KERNEL_NAMES)
@pytest.mark.parametrize("dt", [np.float64, np.float32], ids=["f64", "f32"])
def test_hip_matches_oracle_ragged_sizes(O, name, dt):
    """Sizes around every internal boundary: wave (64), workgroup (256), 2 targets per lane (512), LDS tile (256),
    source splits.  Oracle = CPU restatement on the same inputs."""
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(7)
    tol = 1e-12 if dt == np.float64 else 2e-5
    for (Nt, Ns) in [(1, 1), (1, 700), (63, 257), (255, 256), (257, 255), (513, 1025), (2049, 511), (3000, 4099)]:
        xt, xs, xn, f = _rng_inputs(rng, Nt, Ns, info, dt)
        u = sctl_amd.eval_host(name, xt, xs, xn, f, ctx=ctx_for(name))
        ref = O.eval(name, xt, xs, xn, f, ctx=ctx_for(name))
        assert rel_l2(u, ref) <= tol, (Nt, Ns, rel_l2(u, ref))


def test_empty_inputs_and_accumulate_semantics(O):
    name = "Stokes3D-FxU"
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(3)
    xt, xs, xn, f = _rng_inputs(rng, 100, 50, info, np.float64)
    # no sources: output untouched (accumulating zero); no targets: empty output
    v0 = rng.random(300)
    v = sctl_amd.eval_host(name, xt, np.zeros(0), None, np.zeros(0), v_trg=v0.copy())
    assert np.array_equal(v, v0)
    assert sctl_amd.eval_host(name, np.zeros(0), xs, None, f).size == 0
    # right-sized output is accumulated into; wrong-sized is replaced by a zeroed one (generic-kernel.txx:98-101)
    u = sctl_amd.eval_host(name, xt, xs, None, f)
    u2 = sctl_amd.eval_host(name, xt, xs, None, f, v_trg=v0.copy())
    assert rel_l2(u2 - v0, u) < 1e-13
    u3 = sctl_amd.eval_host(name, xt, xs, None, f, v_trg=np.ones(7))
    assert np.array_equal(u3, u)
    # twice the same call accumulates twice
    u4 = sctl_amd.eval_host(name, xt, xs, None, f, v_trg=u.copy())
    assert rel_l2(u4, 2 * u) < 1e-15


@pytest.mark.parametrize("name", ["Laplace3D-FxU", "Laplace3D-FxdU", "Stokes3D-DxU", "Stokes3D-FxT"])
def test_digits_accuracy_ladder(O, name):
    """digits requests at least that many digits (generic-kernel.txx:46-77): seed / Newton / Halley refinement."""
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(11)
    xt, xs, xn, f = _rng_inputs(rng, 700, 1500, info, np.float64)
    exact = O.eval(name, xt, xs, xn, f)
    errs = {}
    for d in (3, 7, 10, 14, 15, -1, 20):
        errs[d] = rel_l2(sctl_amd.eval_host(name, xt, xs, xn, f, digits=d), exact)
        assert errs[d] <= (10.0 * 10.0 ** (-d) if 0 <= d < 15 else 1e-13), (d, errs[d])
    assert errs[-1] <= 5e-15 and errs[14] <= 1e-13 and errs[7] <= 1e-6


def test_coincident_points_contribute_zero(O):
    """targets == sources: the r = 0 pair is masked to exactly 0 (kernel_functions.hpp:28), nothing becomes NaN/inf."""
    rng = np.random.default_rng(5)
    for name in sctl_amd.KERNEL_NAMES:
        info = sctl_amd.kernel_info(name)
        for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
            xt, xs, xn, f = _rng_inputs(rng, 600, 600, info, dt)
            u = sctl_amd.eval_host(name, xs, xs, xn, f, ctx=ctx_for(name))
            assert np.all(np.isfinite(u)), name
            assert rel_l2(u, O.eval(name, xs, xs, xn, f, ctx=ctx_for(name))) <= tol, name
    # a single point acting on itself gives exactly zero
    u = sctl_amd.eval_host("Laplace3D-FxU", np.array([.3, .4, .5]), np.array([.3, .4, .5]), None, np.array([2.0]))
    assert u[0] == 0.0


def test_analytic_known_answers():
    xt = np.array([0.0, 0.0, 2.0]); xs = np.zeros(3)
    u = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, np.array([3.0]))
    assert abs(u[0] - 3.0 / (8 * np.pi)) < 1e-16
    u = sctl_amd.eval_host("Stokes3D-FxU", xt, xs, None, np.array([0.0, 0.0, 1.0]))
    assert np.allclose(u, [0, 0, 2.0 / (16 * np.pi)], atol=1e-16)
    # Helmholtz with k = 0 reduces to Laplace
    rng = np.random.default_rng(1)
    xt, xs, f = rng.random(300), rng.random(600), rng.random(400) - 0.5
    uh = sctl_amd.eval_host("Helmholtz3D-FxU", xt, xs, None, f, ctx=np.array([0.0, 0.0]))
    ul_re = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f[0::2].copy())
    ul_im = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f[1::2].copy())
    assert rel_l2(uh[0::2], ul_re) < 1e-14 and rel_l2(uh[1::2], ul_im) < 1e-14


def test_fused_laplace_functor_equals_sum_of_reference_functors():
    """Laplace3D-FDxUdU (new) == FxU + DxU for the potential and FxdU for the single-layer gradient."""
    rng = np.random.default_rng(2)
    Nt, Ns = 500, 900
    xt, xs, xn = rng.random(Nt * 3), rng.random(Ns * 3), rng.random(Ns * 3) - 0.5
    q, mu = rng.random(Ns) - 0.5, rng.random(Ns) - 0.5
    fused = sctl_amd.eval_host("Laplace3D-FDxUdU", xt, xs, xn, np.stack([q, mu], 1).ravel().copy()).reshape(Nt, 4)
    pot = sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, q) + sctl_amd.eval_host("Laplace3D-DxU", xt, xs, xn, mu)
    assert rel_l2(fused[:, 0], pot) < 1e-13
    only_q = sctl_amd.eval_host("Laplace3D-FDxUdU", xt, xs, xn, np.stack([q, 0 * mu], 1).ravel().copy()).reshape(Nt, 4)
    grad_sl = sctl_amd.eval_host("Laplace3D-FxdU", xt, xs, None, q).reshape(Nt, 3)
    assert rel_l2(only_q[:, 1:], grad_sl) < 1e-13


def test_device_resident_entry_and_stream(O):
    """torch CUDA tensors through sctl_amd_eval_device on torch's current stream, accumulate semantics on device."""
    import torch
    name = "Laplace3D-DxU"
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(9)
    xt, xs, xn, f = _rng_inputs(rng, 5000, 3000, info, np.float64)
    ref = O.eval(name, xt, xs, xn, f)
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d)
    assert rel_l2(u.cpu().numpy(), ref) < 1e-12
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        u2 = sctl_amd.eval_device(name, *d, v_trg=u.clone())
    side.synchronize()
    assert rel_l2(u2.cpu().numpy(), 2 * ref) < 1e-12
    M = sctl_amd.kernel_matrix_device(name, d[0][:30], d[1][:60], d[2][:60])
    assert rel_l2(M.cpu().numpy(), O.kernel_matrix(name, xt[:30], xs[:60], xn[:60])) < 1e-12


def test_multi_device_entry_slab_partition(O):
    """sctl_amd_eval_host_multi: targets block-partitioned over the device list (fmm-wrapper.txx:507), here the same
    GPU listed several times so the slab arithmetic and the host threads are exercised on a one-GPU box."""
    name = "Stokes3D-FxU"
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(13)
    xt, xs, xn, f = _rng_inputs(rng, 1001, 777, info, np.float64)
    ref = O.eval(name, xt, xs, xn, f)
    for devs in ([0], [0, 0], [0, 0, 0]):
        assert rel_l2(sctl_amd.eval_host(name, xt, xs, xn, f, devices=devs), ref) < 1e-12
    with pytest.raises(sctl_amd.api.SctlAmdError):
        sctl_amd.eval_host(name, xt, xs, xn, f, devices=[0, 99])


def test_error_codes():
    with pytest.raises(KeyError):
        sctl_amd.kernel_id("Laplace3D-Nope")
    with pytest.raises(sctl_amd.api.SctlAmdError):   # Helmholtz without its wavenumber
        sctl_amd.eval_host("Helmholtz3D-FxU", np.zeros(3), np.ones(3), None, np.ones(2))
    with pytest.raises(sctl_amd.api.SctlAmdError):   # double layer without normals
        sctl_amd.eval_host("Laplace3D-DxU", np.zeros(3), np.ones(3), None, np.ones(1))


def test_counters_follow_reference_flop_accounting():
    """Profile::IncrementCounter(FLOP, Ns*Nt*FLOPS()) (generic-kernel.txx:188)."""
    sctl_amd.reset_counters()
    sctl_amd.eval_host("Stokes3D-DxU", np.random.rand(30), np.random.rand(60), np.random.rand(60), np.random.rand(60))
    c = sctl_amd.counters()
    assert c["pair_interactions"] == 200 and c["sctl_flops"] == 200 * 26


def test_device_resident_operator_handle(O):
    """sctl_amd_op_*: coordinates uploaded once; evaluations with new densities, accumulate / overwrite, new sources, and
    the target slabs of a multi-device list (the same GPU listed twice)."""
    name = "Laplace3D-DxU"
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(17)
    xt, xs, xn, f = _rng_inputs(rng, 3001, 2500, info, np.float64)
    for devs in ((0,), (0, 0)):
        op = sctl_amd.DirectOp(name, np.float64, devices=devs)
        op.set_targets(xt)
        op.set_sources(xs, xn)
        u1 = op.eval(f)
        assert rel_l2(u1, O.eval(name, xt, xs, xn, f)) < 1e-12
        f2 = rng.random(f.size) - 0.5
        u2 = op.eval(f2)                                    # overwrite semantics (EvalDirect)
        assert rel_l2(u2, O.eval(name, xt, xs, xn, f2)) < 1e-12
        u3 = op.eval(f, v_trg=u2.copy(), accumulate=True)   # accumulate semantics (GenericKernel::Eval)
        assert rel_l2(u3, O.eval(name, xt, xs, xn, f) + u2) < 1e-12
        xs2, xn2 = rng.random(900 * 3), rng.random(900 * 3) - 0.5
        f3 = rng.random(900) - 0.5
        op.set_sources(xs2, xn2)                            # fewer sources than before: buffers are reused
        assert rel_l2(op.eval(f3), O.eval(name, xt, xs2, xn2, f3)) < 1e-12
        op.set_targets(xt[:300].copy())
        assert rel_l2(op.eval(f3), O.eval(name, xt[:300].copy(), xs2, xn2, f3)) < 1e-12
        op.close()
    with pytest.raises(sctl_amd.api.SctlAmdError):
        sctl_amd.DirectOp(name, np.float64, devices=(0, 99))


def test_reused_host_buffers_are_always_re_read(O):
    """Callers rewrite their coordinate / density arrays in place between evaluations (SCTL keeps one density vector per
    source type).  Every host entry must deliver the NEW contents, however the library moves them (capi.hip: PinnedBuf; round 1 saw
    stale contents here about once per thousand transfers — most likely the memory-pool fault of workspace.hpp)."""
    name = "Stokes3D-DxU"
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(23)
    N = 3000
    xt, xa, na, fa = _rng_inputs(rng, N, N, info, np.float64)
    xb, fb = rng.random(N * 3), rng.random(N * 3) - 0.5
    ref = {(0, 0): O.eval(name, xt, xa, na, fa), (1, 0): O.eval(name, xt, xb, na, fa), (0, 1): O.eval(name, xt, xa, na, fb),
           (1, 1): O.eval(name, xt, xb, na, fb)}
    op = sctl_amd.DirectOp(name)
    op.set_targets(xt)
    buf, fbuf = xa.copy(), fa.copy()
    for it in range(600):
        ub, uf = it & 1, (it >> 1) & 1
        buf[:] = xb if ub else xa
        fbuf[:] = fb if uf else fa
        op.set_sources(buf, na)
        assert rel_l2(op.eval(fbuf), ref[(ub, uf)]) < 1e-12, it
        assert rel_l2(sctl_amd.eval_host(name, xt, buf, na, fbuf), ref[(ub, uf)]) < 1e-12, it
    op.close()


@pytest.mark.parametrize("k", [(7.5, 0.0), (40.0, 0.3), (3.0, 60.0), (5.0, -2.0), (3.0e4, 0.0), (2.0e4, 1.0), (0.0, 4.0), (-12.0, 0.5),
                               # round 3, the one-reduction form (Re k > 0, |Im k| <= Re k / 4): both ends of the decay ratio, just outside it (the
                               # two-reduction form), and 248 whole periods across the cloud (the last entries of the period-factor table)
                               (40.0, 10.0), (40.0, -10.0), (40.0, 10.5), (900.0, 5.0), (1e-3, 2e-4)],
                         ids=lambda k: "k=%g%+gi" % k)
@pytest.mark.parametrize("shape", [(700, 900), (5000, 6000)], ids=lambda s: "%dx%d" % s)
def test_helmholtz_wavenumbers(O, k, shape):
    """The fp64 Helmholtz kernel's table-driven sincos / exp over the wavenumber plane: real k (no exponential), strong
    decay and growth, and |Re k| r beyond the table path's range (the tile is re-run through libm).  The small shape runs
    the careful (masked) pass only, the large one the speculative pass; sources include the targets (r = 0 pairs)."""
    nt, ns = shape
    rng = np.random.default_rng(31)
    xs = rng.random(ns * 3)
    xt = np.concatenate([xs[:3 * 40], rng.random((nt - 40) * 3)])
    f = rng.random(ns * 2) - 0.5
    ctx = np.array(k, dtype=np.float64)
    ref = O.eval("Helmholtz3D-FxU", xt, xs, None, f, ctx=ctx)
    u = sctl_amd.eval_host("Helmholtz3D-FxU", xt, xs, None, f, ctx=ctx)
    assert np.all(np.isfinite(u))
    # |Re k| r of 1e4 amplifies the rounding of r itself (relative 1e-16) to an absolute phase error of 1e-12
    tol = 1e-12 if abs(k[0]) < 1e3 else 1e-10
    assert rel_l2(u, ref) <= tol, rel_l2(u, ref)


@pytest.mark.parametrize("name", ["Laplace3D-FxU", "Stokes3D-DxU", "Laplace3D-FDxUdU", "Helmholtz3D-FxU"])
def test_kernel_matrix_batch_matches_per_block_oracle(O, name):
    """sctl_amd_kernel_matrix_batch_host: many operator blocks in one launch == KernelMatrix per block (generic-kernel.txx:191-307),
    including empty blocks, blocks narrower than one 64-target tile and blocks spanning several tiles."""
    inf = O.info(name)
    rng = np.random.default_rng(3)
    nt = np.array([1, 70, 0, 64, 5, 200, 33], dtype=np.int64)
    ns = np.array([3, 9, 4, 0, 130, 17, 1], dtype=np.int64)
    xt, xs = rng.random(int(nt.sum()) * 3), rng.random(int(ns.sum()) * 3)
    xn = rng.random(int(ns.sum()) * inf["nd"]) - 0.5
    ctx = np.array([7.5, 0.3]) if name.startswith("Helmholtz") else None
    blocks = sctl_amd.api.kernel_matrix_batch_host(name, nt, ns, xt, xs, xn, ctx=ctx)
    to, so = np.concatenate([[0], np.cumsum(nt)]), np.concatenate([[0], np.cumsum(ns)])
    for b in range(nt.size):
        assert blocks[b].shape == (ns[b] * inf["k0"], nt[b] * inf["k1"])
        if blocks[b].size == 0:
            continue
        ref = O.kernel_matrix(name, xt[to[b] * 3:to[b + 1] * 3].copy(), xs[so[b] * 3:so[b + 1] * 3].copy(),
                              xn[so[b] * inf["nd"]:so[b + 1] * inf["nd"]].copy() if inf["nd"] else None, ctx=ctx)
        assert rel_l2(blocks[b], ref) <= 1e-13, (b, rel_l2(blocks[b], ref))
    # fp32
    b32 = sctl_amd.api.kernel_matrix_batch_host(name, nt, ns, xt.astype(np.float32), xs.astype(np.float32), xn.astype(np.float32), ctx=ctx)
    assert rel_l2(b32[5].astype(np.float64), blocks[5]) <= 1e-5


def test_host_entries_are_reentrant(O):
    """The reference calls KernelMatrix inside `omp parallel for` (boundary_integral.txx:949-986) and Eval from whatever thread
    the caller is on: concurrent host-pointer calls from several threads (each gets its own stream, device buffers and pinned
    staging; the scratch arena is keyed by stream) must all be right."""
    import threading
    rng = np.random.default_rng(23)
    jobs, results, errors = [], {}, []
    for i in range(8):
        name = ("Stokes3D-DxU", "Laplace3D-FxU", "Helmholtz3D-FxU", "Laplace3D-FxdU")[i % 4]
        info = sctl_amd.kernel_info(name)
        nt, ns = 700 + 97 * i, 3000 + 501 * i                       # several source splits: every call uses the scratch arena
        xt, xs, xn, f = _rng_inputs(rng, nt, ns, info, np.float64)
        jobs.append((i, name, xt, xs, xn, f, np.array([7.5, 0.3]) if name.startswith("Helm") else None))

    def work(job):
        i, name, xt, xs, xn, f, ctx = job
        try:
            for rep in range(5):
                u = sctl_amd.eval_host(name, xt, xs, xn, f, ctx=ctx)
                M = sctl_amd.kernel_matrix_host(name, xt[:60].copy(), xs[:90].copy(), xn[:30 * sctl_amd.kernel_info(name)["nd"]].copy() if xn.size else xn, ctx=ctx)
                results[(i, rep)] = (u, M)
        except Exception as e:                                      # noqa: BLE001 - reported below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, name, xt, xs, xn, f, ctx in jobs:
        ref = O.eval(name, xt, xs, xn, f, ctx=ctx)
        refM = O.kernel_matrix(name, xt[:60].copy(), xs[:90].copy(), xn[:30 * O.info(name)["nd"]].copy() if xn.size else xn, ctx=ctx)
        for rep in range(5):
            u, M = results[(i, rep)]
            assert rel_l2(u, ref) <= 1e-12, (i, name, rep, rel_l2(u, ref))
            assert rel_l2(M, refM) <= 1e-12, (i, name, rep)
    sctl_amd.api.trim()                                              # releases the cached scratch blocks; the library keeps working
    i, name, xt, xs, xn, f, ctx = jobs[0]
    assert rel_l2(sctl_amd.eval_host(name, xt, xs, xn, f, ctx=ctx), O.eval(name, xt, xs, xn, f, ctx=ctx)) <= 1e-12


@pytest.mark.parametrize("devs", [(0,), (0, 0)])
def test_operator_handle_applies_weights_and_target_normals_on_the_device(O, devs):
    """sctl_amd_op_set_source_weights / _target_normals (the far-field pre/post steps of boundary_integral.txx:1040-1071 on the device):
    u[t][k] = sum_l (sum_s K(x_t - x_s)[.][k][l] w_s f_s) n_t[l], also with Morton-ordered slabs on several devices, and the
    setters' lifetime rules (new sources drop the weights, new targets drop the normals)."""
    name = "Stokes3D-FxT"                                  # 3 -> 9 output components, contracted to 3
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(29)
    nt, ns = 1500, 2100
    xt, xs, xn, f = _rng_inputs(rng, nt, ns, info, np.float64)
    w, ntrg = rng.random(ns) * 0.01, rng.random(nt * 3) - 0.5
    full = O.eval(name, xt, xs, xn, (f.reshape(ns, 3) * w[:, None]).ravel())
    want = (full.reshape(nt, 3, 3) * ntrg.reshape(nt, 1, 3)).sum(-1).ravel()
    op = sctl_amd.DirectOp(name, np.float64, devices=devs)
    op.set_targets(xt)
    op.set_sources(xs, xn)
    op.set_source_weights(w)
    op.set_target_normals(ntrg)
    u = op.eval(f)
    assert u.size == nt * 3 and rel_l2(u, want) <= 1e-12, rel_l2(u, want)
    u2 = op.eval(f, v_trg=u.copy(), accumulate=True)
    assert rel_l2(u2, 2 * want) <= 1e-12
    op.set_target_normals(None)                            # cleared: the full 9 components again, still weighted
    assert rel_l2(op.eval(f), full) <= 1e-12
    op.set_sources(xs, xn)                                 # new sources drop the weights
    assert rel_l2(op.eval(f), O.eval(name, xt, xs, xn, f)) <= 1e-12
    with pytest.raises(sctl_amd.api.SctlAmdError):         # nothing to contract for a scalar kernel
        lap = sctl_amd.DirectOp("Laplace3D-FxU")
        lap.set_targets(xt)
        lap.set_target_normals(ntrg)


def test_host_entries_leave_the_callers_current_device_alone(O):
    """Host-pointer entries pick their device for the duration of the call only: a caller's HIP state (here torch's) is unchanged.
    On a one-GPU box the check is that nothing breaks and torch still computes on its device afterwards."""
    import torch
    torch.cuda.set_device(0)
    before = torch.cuda.current_device()
    rng = np.random.default_rng(1)
    info = sctl_amd.kernel_info("Laplace3D-DxU")
    xt, xs, xn, f = _rng_inputs(rng, 500, 700, info, np.float64)
    u = sctl_amd.eval_host("Laplace3D-DxU", xt, xs, xn, f)
    op = sctl_amd.DirectOp("Laplace3D-DxU")
    op.set_targets(xt); op.set_sources(xs, xn)
    u2 = op.eval(f)
    op.close()
    sctl_amd.kernel_matrix_host("Laplace3D-DxU", xt[:30].copy(), xs[:60].copy(), xn[:60].copy())
    assert torch.cuda.current_device() == before
    assert float(torch.ones(8, device="cuda").sum()) == 8.0
    assert rel_l2(u, u2) <= 1e-14


def test_init_and_finalize_are_optional_and_repeatable(O):
    """sctl_amd_init touches every GPU and reports their number; sctl_amd_finalize gives back the cached operators, scratch blocks and this
    thread's streams / buffers / staging — and the library goes on working (everything initialises lazily)."""
    import torch
    assert sctl_amd.init() == torch.cuda.device_count() == sctl_amd.device_count()
    rng = np.random.default_rng(12)
    xt, xs, f = rng.random(3000 * 3), rng.random(5000 * 3), rng.random(5000 * 3) - 0.5
    ref = O.eval("Stokes3D-FxU", xt, xs, None, f)
    for devices in ([0], [0, 0, 0]):
        for _ in range(2):
            assert rel_l2(sctl_amd.eval_host("Stokes3D-FxU", xt, xs, None, f, devices=devices), ref) < 1e-12
            sctl_amd.finalize()
    assert sctl_amd.init() == torch.cuda.device_count()


@pytest.mark.parametrize("dt", [np.float64, np.float32], ids=["f64", "f32"])
def test_every_kernel_gives_bit_identical_results_run_to_run(dt):
    """No sum is accumulated with atomics and every reduction runs in a fixed order, so repeated evaluations must agree to the last bit — through the exact
    all-pairs kernels (with source splits and their reduction), KernelMatrix and the batched list kernel, with other kernels (copies, fills) between the
    launches.  The counterpart for the tile-centred kernels, where round 3 found a timing fault of compiled code, is in test_gpu_centered.py."""
    import torch
    from sctl_amd.lists import grid_neighbour_lists, points_in_boxes
    rng = np.random.default_rng(17)
    Nt, Ns = 20000, 33000
    bits = np.int32 if dt == np.float32 else np.int64
    cnt = rng.integers(1, 90, 6 ** 3)
    xl = points_in_boxes(6, cnt, rng).astype(dt)
    lists = grid_neighbour_lists(6, cnt, cnt)
    for name in sctl_amd.KERNEL_NAMES:
        info = sctl_amd.kernel_info(name)
        ctx = ctx_for(name)
        xt, xs = rng.random(Nt * 3).astype(dt), rng.random(Ns * 3).astype(dt)
        xn = (rng.random(Ns * info["nd"]) - 0.5).astype(dt) if info["nd"] else None
        f = (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
        d = [None if a is None else torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
        runs = [sctl_amd.eval_device(name, *d, ctx=ctx).clone() for _ in range(3)]
        tb = torch.int32 if dt == np.float32 else torch.int64
        assert bool(torch.isfinite(runs[0]).all())
        for r in runs[1:]:
            assert int((r.view(tb) != runs[0].view(tb)).sum()) == 0, ("eval_device", name)
        m = [sctl_amd.kernel_matrix_host(name, xt[:600], xs[:300], None if xn is None else xn[:100 * info["nd"]], ctx=ctx) for _ in range(2)]
        assert np.array_equal(m[0].view(bits), m[1].view(bits)), ("kernel_matrix", name)
        fl = (rng.random(int(cnt.sum()) * info["k0"]) - 0.5).astype(dt)
        nl = (rng.random(int(cnt.sum()) * info["nd"]) - 0.5).astype(dt) if info["nd"] else None
        u = [sctl_amd.eval_lists_host(name, *lists, xl, xl, nl, fl, ctx=ctx) for _ in range(2)]
        assert np.all(np.isfinite(u[0])) and np.array_equal(u[0].view(bits), u[1].view(bits)), ("lists", name)
