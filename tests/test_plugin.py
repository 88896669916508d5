"""The device functor surface (reference doc/tutorial/kernels.rst:11-84, generic-kernel.hpp:33-52): a kernel the
library does not contain — Yukawa, exp(-lambda r)/(4 pi r), tests/plugin/yukawa_kernel.hip — is compiled by hipcc against the
installed device headers (include/sctl_amd/device/kernel_plugin.hpp) into its own shared object, registered at load, and then served
by every entry of the C ABI.  Expected values: the same functor on the REAL reference's GenericKernel (tests/golden/Yukawa3D-FxU.npz,
oracle/gen_golden_plugin.py) and plain numpy.  CPU: the plugin builds (gfx950 cross-compile), loads and registers; bad descriptors
are refused.  GPU: it computes."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import sctl_amd
from conftest import GOLDEN, ROOT, rel_l2
from sctl_amd.rand48 import Rand48, point_cloud

NAME = "Yukawa3D-FxU"
_G = np.load(os.path.join(GOLDEN, NAME + ".npz"))
MAN = json.loads(bytes(_G["manifest"]).decode())
LAM = MAN["lam"]


@pytest.fixture(scope="module")
def plugin(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("plugin") / "libyukawa_kernel.so")
    libdir = os.path.join(ROOT, "sctl_amd")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "plugin", "yukawa_kernel.hip"), "-o", so, "-L" + libdir, "-lsctl_amd", "-Wl,-rpath," + libdir], check=True)
    try:
        sctl_amd.kernel_id(NAME)
    except KeyError:
        assert sctl_amd.load_plugin(so) == [NAME]
    return so


def numpy_yukawa(xt, xs, f, lam):
    d = xt.reshape(-1, 1, 3).astype(np.float64) - xs.reshape(1, -1, 3).astype(np.float64)
    r = np.sqrt((d * d).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        G = np.where(r > 0, np.exp(-lam * r) / (4 * np.pi * r), 0.0)
    return G @ f.astype(np.float64)


def test_plugin_builds_loads_and_registers(plugin):
    info = sctl_amd.kernel_info(NAME)
    assert info["id"] >= len(sctl_amd.KERNEL_NAMES) and (info["k0"], info["k1"], info["nd"], info["flops"], info["ctx_bytes"]) == (1, 1, 0, 10, 8)
    assert abs(info["scale"] - 1 / (4 * np.pi)) < 1e-17 and sctl_amd.flops_per_pair(NAME) == 3 + 10 + 2
    L = sctl_amd.lib()
    assert L.sctl_amd_num_kernels() > len(sctl_amd.KERNEL_NAMES) and L.sctl_amd_kernel_name(info["id"]) == NAME.encode()
    assert sctl_amd.load_plugin(plugin) == []                      # loading the same object again registers nothing new
    assert sctl_amd.plan(NAME, 0, 1 << 20, 1 << 20)["path"] == "exact"
    # the context is checked like a built-in kernel's
    z = np.zeros(3)
    p = z.ctypes.data_as(C.c_void_p)
    assert L.sctl_amd_eval_host(info["id"], 0, 1, 1, p, p, None, p, p, -1, None, 0, 0) == -5


def test_register_refuses_foreign_or_inconsistent_descriptors(plugin):
    class Desc(C.Structure):
        _fields_ = [("abi_version", C.c_int), ("desc_bytes", C.c_int), ("entry_bytes", C.c_int), ("src_dim", C.c_int), ("trg_dim", C.c_int), ("normal_dim", C.c_int),
                    ("flops", C.c_int), ("ctx_bytes", C.c_int), ("scale", C.c_double), ("name", C.c_char_p), ("launch_table", C.c_void_p)]
    L = sctl_amd.lib()
    L.sctl_amd_register_kernel.argtypes = [C.POINTER(Desc)]
    n0 = L.sctl_amd_num_kernels()
    assert L.sctl_amd_register_kernel(None) == -2
    d = Desc(abi_version=1, desc_bytes=C.sizeof(Desc), entry_bytes=8, name=b"Other", launch_table=None)
    assert L.sctl_amd_register_kernel(C.byref(d)) == -2 and b"other device headers" in L.sctl_amd_last_error()
    assert L.sctl_amd_num_kernels() == n0
    assert L.sctl_amd_load_plugin(b"/nonexistent/libplugin.so") == -2


def test_load_plugin_reports_an_object_that_registers_nothing(tmp_path):
    """A shared object that loads but adds no kernel (no SCTL_AMD_REGISTER_KERNEL in it, or every registration refused) is an ERROR
    carrying the reason — not a silent 0."""
    import subprocess
    src = tmp_path / "empty.c"
    src.write_text("int sctl_amd_test_nothing_here = 1;\n")
    so = str(tmp_path / "libempty_plugin.so")
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", so], check=True)
    L = sctl_amd.lib()
    assert L.sctl_amd_load_plugin(so.encode()) == -2
    assert b"registered no kernel" in L.sctl_amd_last_error()
    from sctl_amd.api import SctlAmdError
    with pytest.raises(SctlAmdError):
        sctl_amd.load_plugin(so)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN["cases"], ids=lambda c: c["key"])
def test_plugin_kernel_matches_reference_functor(plugin, case):
    dt = np.float64 if case["dtype"] == "f64" else np.float32
    xt, xs, xn, f = point_cloud(case["seed"], case["Nt"], case["Ns"], 1, 0, dt)
    if case["self_targets"]:
        xt = xs
    ctx = np.array([LAM])
    u = sctl_amd.eval_host(NAME, xt, xs, None, f, digits=case["digits"], ctx=ctx)
    tol = 10.0 * 10.0 ** -case["digits"] if case["digits"] >= 0 else (1e-12 if dt == np.float64 else 2e-5)
    assert np.all(np.isfinite(u)) and rel_l2(u, _G[case["key"]]) <= tol                 # the reference's GenericKernel on the same functor
    assert rel_l2(u, numpy_yukawa(xt, xs, f, LAM)) <= tol                                  # independent mathematics


@pytest.mark.gpu
def test_plugin_kernel_through_every_entry(plugin):
    import torch
    rng = np.random.default_rng(12)
    Nt, Ns = 70000, 5000                      # enough targets for the source-split / two-targets-per-lane plans
    xt, xs, f = rng.random(Nt * 3), rng.random(Ns * 3), rng.random(Ns) - 0.5
    ctx = np.array([LAM])
    sel = np.arange(0, Nt, 97)
    ref = numpy_yukawa(xt.reshape(-1, 3)[sel].ravel(), xs, f, LAM)
    u = sctl_amd.eval_host(NAME, xt, xs, None, f, ctx=ctx)
    assert rel_l2(u[sel], ref) < 1e-13
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, f)]
    ud = sctl_amd.eval_device(NAME, d[0], d[1], None, d[2], ctx=ctx)                       # device-resident entry
    assert rel_l2(ud.cpu().numpy(), u) < 1e-15
    op = sctl_amd.DirectOp(NAME, np.float64, ctx=ctx)                                      # operator handle (ParticleFMM's resident form)
    op.set_targets(xt); op.set_sources(xs)
    assert rel_l2(op.eval(f), u) < 1e-15
    op.close()
    um = sctl_amd.eval_host(NAME, xt, xs, None, f, ctx=ctx, devices=[0, 0])                # one-process multi-device slabs
    assert rel_l2(um, u) < 1e-15
    xtm, xsm, _, _ = point_cloud(MAN["matrix"]["seed"], MAN["matrix"]["Nt"], MAN["matrix"]["Ns"], 1, 0, np.float64)
    M = sctl_amd.kernel_matrix_host(NAME, xtm, xsm, None, ctx=ctx)                         # KernelMatrix vs the reference's
    assert rel_l2(M, _G["matrix"]) < 1e-12
    Mb = sctl_amd.kernel_matrix_batch_host(NAME, [40, 0, 17], [30, 5, 12], np.concatenate([xtm, xt[:51]]), np.concatenate([xsm, xs[:51]]), None, ctx=ctx)
    assert rel_l2(Mb[0], _G["matrix"]) < 1e-12 and Mb[2].shape == (12, 17)
    one = [np.array([v], dtype=np.int64) for v in (0, 1000, 0, Ns)]                        # list evaluation
    ul = sctl_amd.eval_lists_host(NAME, *one, xt[:3000], xs, None, f, ctx=ctx)
    assert rel_l2(ul, u[:1000]) < 1e-14
    u32 = sctl_amd.eval_host(NAME, xt[:30000].astype(np.float32), xs.astype(np.float32), None, f.astype(np.float32), ctx=ctx)
    assert rel_l2(u32, u[:10000]) < 3e-5


@pytest.mark.gpu
def test_plugin_kernel_from_the_cpp_host_surface(plugin, tmp_path):
    """GenericKernel<descriptor> + ParticleFMM of include/sctl_amd with a functor declared by SCTL_AMD_UKERNEL (tests/cpp/plugin_driver.cpp)."""
    from test_cpp_host import _build, _read_vector
    exe = _build(tmp_path, "plugin_driver")
    out = str(tmp_path / "u.bin")
    N = 2000
    p = subprocess.run([exe, plugin, str(N), out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    g = Rand48(0)
    xt, xs, f = g.drand48(N * 3) - 0.5, g.drand48(N * 3) - 0.5, g.drand48(N) - 0.5
    assert rel_l2(_read_vector(out), numpy_yukawa(xt, xs, f, 2.5)) < 1e-13
    M = _read_vector(out + ".mat").reshape(7, 5)
    d = xt[:15].reshape(1, 5, 3) - xs[:21].reshape(7, 1, 3)
    r = np.sqrt((d * d).sum(-1))
    assert rel_l2(M, np.exp(-2.5 * r) / (4 * np.pi * r)) < 1e-13
