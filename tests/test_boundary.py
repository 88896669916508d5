"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/sctl_amd.h
declares, argument errors are reported (no compute happens without a GPU), the launch planner is sane, and the
product tree never touches oracle/."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import sctl_amd
from conftest import ROOT
from sctl_amd import api


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "sctl_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sctl_amd_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert declared == sorted(api.SYMBOLS)
    L = sctl_amd.lib()
    for sym in declared:
        assert hasattr(L, sym), sym
    # and nothing C++-mangled leaks as the public surface: the declared names resolve as plain C symbols
    out = subprocess.run(["nm", "-D", "--defined-only", sctl_amd.library_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sctl_amd_[a-z_0-9]+)$", out, flags=re.M))
    assert set(declared) <= exported


def test_version_and_registry():
    L = sctl_amd.lib()
    assert L.sctl_amd_version() >= 100
    assert L.sctl_amd_kernel_id(b"Laplace3D-FxU") == 0
    assert L.sctl_amd_kernel_id(b"NoSuchKernel") < 0           # the "is it supported" query
    assert L.sctl_amd_kernel_name(99) is None
    for i, name in enumerate(sctl_amd.KERNEL_NAMES):
        assert sctl_amd.kernel_info(name)["id"] == i
    assert sctl_amd.kernel_info("Helmholtz3D-FxU")["ctx_bytes"] == 16
    with pytest.raises(KeyError):
        sctl_amd.kernel_id("Laplace3D-SL")


def test_argument_errors_before_any_device_work():
    L = sctl_amd.lib()
    z = np.zeros(3)
    p = z.ctypes.data_as(ctypes.c_void_p)
    # unknown kernel, bad precision tag, negative size, missing normals, missing context
    assert L.sctl_amd_eval_host(99, 0, 1, 1, p, p, None, p, p, -1, None, 0, 0) == -1
    assert L.sctl_amd_eval_host(0, 7, 1, 1, p, p, None, p, p, -1, None, 0, 0) == -2
    assert L.sctl_amd_eval_host(0, 0, -1, 1, p, p, None, p, p, -1, None, 0, 0) == -2
    assert L.sctl_amd_eval_host(1, 0, 1, 1, p, p, None, p, p, -1, None, 0, 0) == -2
    assert b"normals" in L.sctl_amd_last_error()
    assert L.sctl_amd_eval_host(9, 0, 1, 1, p, p, None, p, p, -1, None, 0, 0) == -5
    assert L.sctl_amd_kernel_matrix_host(99, 0, 1, 1, p, p, None, p, -1, None, 0, 0) == -1


def test_no_cpu_fallback_without_device():
    if sctl_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.SctlAmdError, match="no HIP device"):
        sctl_amd.eval_host("Laplace3D-FxU", np.zeros(3), np.ones(3), None, np.ones(1))
    with pytest.raises(api.SctlAmdError, match="no HIP device"):
        sctl_amd.kernel_matrix_host("Laplace3D-FxU", np.zeros(3), np.ones(3), None)


def test_launch_planner():
    """Enough workgroups for 256 CUs at every BASELINE size; source splits only when targets are few."""
    for name in sctl_amd.KERNEL_NAMES:
        for real in (0, 1):
            for N in (1, 300, 1 << 14, 1 << 18, 1 << 20, 1 << 23):
                p = sctl_amd.plan(name, real, N, N)
                assert p["trg_per_lane"] in (1, 2, 3, 4) and p["src_splits"] >= 1
                if p["path"] == "exact":
                    assert p["workgroups"] >= min(1024, ((N + 255) // 256) * ((N + 255) // 256))
                k1 = sctl_amd.kernel_info(name)["k1"]
                assert p["workspace_bytes"] == (0 if p["src_splits"] == 1 else p["src_splits"] * N * k1 * (8 if real == 0 else 4))
    # the exact kernel from 2^34 pairs on: a split's source data <= 2 MB, splits in eights (one share per XCD), <= 64 splits, <= 4 GB of partial sums
    for name, real, logn in (("Stokes3D-FxU", 0, 20), ("Stokes3D-FxU", 0, 18), ("Laplace3D-FDxUdU", 0, 20), ("Helmholtz3D-FxU", 0, 20), ("Helmholtz3D-FxU", 1, 23),
                             ("Laplace3D-FxdU", 0, 17)):
        p, i = sctl_amd.plan(name, real, 1 << logn, 1 << logn), sctl_amd.kernel_info(name)
        rs = 8 if real == 0 else 4
        assert p["path"] == "exact" and p["trg_per_lane"] == 2 and p["src_splits"] % 8 == 0 and 8 <= p["src_splits"] <= 64, (name, p)
        assert p["workspace_bytes"] <= 4 << 30
        assert ((rs * (3 + i["nd"] + i["k0"]) << logn) / p["src_splits"] <= (2 << 20) or p["src_splits"] == 64 or
                (p["src_splits"] + 8) * (i["k1"] * rs << logn) > 4 << 30), (name, p)
    assert sctl_amd.plan("Stokes3D-FxU", 0, 1 << 20, 1 << 20)["src_splits"] == 24
    assert sctl_amd.plan("Stokes3D-FxU", 0, 1 << 16, 1 << 16)["src_splits"] == 16          # under 2^34 pairs the old rule: 8 workgroups per CU, no more
    small = sctl_amd.plan("Laplace3D-FxU", 0, 1 << 14, 1 << 14)
    assert small["src_splits"] == 16 and small["path"] == "exact"
    if os.environ.get("SCTL_AMD_CENTERED") != "0":      # the headline problem takes the tile-centred Laplace path: one wave per workgroup
        head = sctl_amd.plan("Laplace3D-FxU", 0, 1 << 20, 1 << 20)
        assert head["path"] == "tile-centred" and head["src_splits"] == 32 and head["workgroups"] == 4096 * 32 and head["trg_per_lane"] == 4   # 256 targets per wave (fp64)
        assert sctl_amd.plan("Laplace3D-FxU", 1, 1 << 20, 1 << 20)["path"] == "tile-centred"
        assert sctl_amd.plan("Laplace3D-DxU", 0, 1 << 20, 1 << 20)["path"] == "tile-centred"  # scalar Laplace kernels have a centred form
        grad = sctl_amd.plan("Laplace3D-FxdU", 0, 1 << 20, 1 << 20)                           # the gradient: far sources as moments (round 4), three targets per lane, fp64 only
        assert grad["path"] == "tile-centred" and grad["trg_per_lane"] == 3 and grad["workspace_bytes"] == grad["src_splits"] * 3 * 8 << 20
        assert sctl_amd.plan("Laplace3D-FxdU", 1, 1 << 20, 1 << 20, digits=9)["path"] == "exact"   # (fp32 at the seed's accuracy: the matrix-core moments kernel, below)
        up = sctl_amd.plan("Stokes3D-FxUP", 0, 1 << 20, 1 << 20)                              # velocity + pressure: the pressure IS one of the Stokeslet's four far moments
        assert up["path"] == "tile-centred" and up["trg_per_lane"] == 4 and up["workspace_bytes"] == up["src_splits"] * 4 * 8 << 20
        for name in ("Stokes3D-FxU", "Stokes3D-FSxU"):                                        # ... for the Stokeslet itself the moments save one instruction of 21: not taken
            assert sctl_amd.plan(name, 0, 1 << 20, 1 << 20)["path"] == "exact"
        assert sctl_amd.plan("Laplace3D-FDxUdU", 0, 1 << 20, 1 << 20)["path"] == "exact"      # its centred form measured slower than the exact kernel
        # which units run the far pairs (sctl_amd_eval_pipe): the bf16 matrix cores only for fp32 scalar Laplace at the seed's accuracy
        if os.environ.get("SCTL_AMD_MFMA_F32") != "0":
            for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
                assert sctl_amd.plan(name, 1, 1 << 20, 1 << 20)["pipe"].startswith("bf16 matrix cores")
                assert sctl_amd.plan(name, 1, 1 << 20, 1 << 20, digits=9)["pipe"] == "vector pipe"
                assert sctl_amd.plan(name, 0, 1 << 20, 1 << 20)["pipe"] == "vector pipe"
        assert small["pipe"] == "vector pipe" and sctl_amd.plan("Helmholtz3D-FxU", 1, 1 << 20, 1 << 20)["pipe"] == "vector pipe"
        # ... and for seven fp32 kernels with several outputs per target (round 4: r.f, r.n are further contractions against the same targets' operand), at the seed's accuracy only
        if os.environ.get("SCTL_AMD_MFMA_F32") != "0":
            for name in ("Stokes3D-FxU", "Stokes3D-FSxU", "Stokes3D-FxUP", "Stokes3D-DxU", "Stokes3D-FxT", "Laplace3D-FxdU", "Laplace3D-FDxUdU"):
                p32 = sctl_amd.plan(name, 1, 1 << 20, 1 << 20)
                assert p32["path"] == "tile-centred" and p32["pipe"].startswith("bf16 matrix cores") and p32["trg_per_lane"] == 2, p32
                p9 = sctl_amd.plan(name, 1, 1 << 20, 1 << 20, digits=9)
                assert p9["path"] == "exact" and p9["pipe"] == "vector pipe" and p9["trg_per_lane"] == 2 and p9["src_splits"] % 8 == 0, p9
        # a split's source data fits half an XCD's L2 (2 MB) and the splits come in eighths, one share per XCD (centered.hip)
        for name, real, logn in (("Laplace3D-FxU", 1, 23), ("Laplace3D-FxU", 1, 21), ("Laplace3D-DxU", 0, 20), ("Laplace3D-FxU", 0, 21)):
            p, i = sctl_amd.plan(name, real, 1 << logn, 1 << logn), sctl_amd.kernel_info(name)
            per_source = (8 if real == 0 else 4) * (3 + i["nd"] + i["k0"])
            assert p["src_splits"] % 8 == 0 and p["src_splits"] <= 64
            assert (per_source << logn) / p["src_splits"] <= (2 << 20) or p["src_splits"] == 64


def test_product_tree_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/ (the checker is not the product)."""
    offenders = []
    for base in ("sctl_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for fn in files:
                if fn.endswith((".py", ".hpp", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b|libsctl_oracle|sctl_oracle_|oracle/_ref|libsctl_ref", txt, flags=re.M):
                        offenders.append(os.path.join(dp, fn))
    assert offenders == []
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("import oracle") == 1 and "def cpu_baseline" in bench


def test_rand48_matches_posix():
    """Known drand48 values after srand48(7) (checked against glibc when the fixtures were generated)."""
    from sctl_amd.rand48 import Rand48
    v = Rand48(7).drand48(3)
    assert v[0] == 0.2664441967654092 and v[1] == 0.68203523019062118 and v[2] == 0.26549059342699977
    a = Rand48(3).drand48(20000)          # block jump-ahead path == sequential path
    g = Rand48(3)
    b = np.concatenate([g.drand48(100) for _ in range(200)])
    assert np.array_equal(a, b)


def test_reference_side_binding_compiles_against_the_real_reference(oracle_mod):
    """include/sctl_amd/sctl_dropin.hpp (HipKernel<uKernel> : sctl::GenericKernel<uKernel>) was compiled with the reference's own
    ParticleFMM / BoundaryIntegralOp by the build (oracle/Makefile `ref`); the resulting shim loads next to libsctl_amd.so and reports
    the reference's kernel table.  (What it computes is checked on the GPU: tests/test_gpu_dropin.py.)"""
    if not os.path.isdir("/root/reference/include/sctl") and oracle_mod.dropin() is None:
        pytest.skip("no reference tree where the build ran: nothing to compile against")
    sctl_amd.lib()
    D = oracle_mod.dropin()
    assert D is not None, "oracle/_ref/libsctl_ref_dropin.so missing: run `make -C oracle ref` after building libsctl_amd.so"
    for name in sctl_amd.KERNEL_NAMES:
        a, b = D.info(name), sctl_amd.kernel_info(name)
        assert (a["k0"], a["k1"], a["nd"], a["flops"]) == (b["k0"], b["k1"], b["nd"], b["flops"]) and abs(a["scale"] - b["scale"]) < 1e-15
    out = subprocess.run(["ldd", os.path.join(ROOT, "oracle", "_ref", "libsctl_ref_dropin.so")], capture_output=True, text=True).stdout
    assert "libsctl_amd.so" in out


def test_dropin_device_list_from_the_environment(oracle_mod):
    """sctl_dropin.hpp reads its process-wide device list once: SCTL_AMD_DEVICES (a list or `all`), else the launcher's node-local rank, else every
    visible GPU; SCTL_AMD_MIN_PAIRS_PER_DEVICE sets the work-per-GPU threshold.  (One subprocess per environment: the list is a static.)"""
    if oracle_mod.dropin() is None:
        pytest.skip("oracle/_ref/libsctl_ref_dropin.so was not built (no reference tree where build() ran)")
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import sctl_amd, oracle; sctl_amd.lib(); print(oracle.dropin().get_devices(), sctl_amd.device_count())" % ROOT)
    drop = ("SCTL_AMD_DEVICES", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK", "SCTL_AMD_MIN_PAIRS_PER_DEVICE")
    base = {k: v for k, v in os.environ.items() if k not in drop}

    def run(**extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        (devs, min_pairs), n_gpu = eval(r.stdout.strip().splitlines()[-1].replace(") ", "), ", 1))
        return devs, min_pairs, n_gpu

    devs, min_pairs, n_gpu = run()
    assert devs == list(range(max(n_gpu, 1))) and min_pairs == 1 << 32
    assert run(SCTL_AMD_DEVICES="0,2,3")[0] == [0, 2, 3]
    assert run(SCTL_AMD_DEVICES="all")[0] == list(range(max(n_gpu, 1)))
    assert run(SCTL_AMD_DEVICES="1", SCTL_AMD_MIN_PAIRS_PER_DEVICE="12345")[:2] == ([1], 12345)
    assert run(SLURM_LOCALID="5")[0] == [5 % n_gpu if n_gpu else 0]                 # one rank per GPU under a launcher
    assert run(LOCAL_RANK="3", SCTL_AMD_DEVICES="0,1")[0] == [0, 1]                 # the explicit list wins


def test_device_assembly_of_the_matrix_core_kernels_keeps_mfma_operands_untouched():
    """tools/check_mfma_operands.py on the assembly hipcc makes of sctl_amd/csrc/centered.hip with the library's own flags: no write to a v_mfma's A / B
    registers within 24 instructions of its issue (a precaution the compiler does not take by itself: DESIGN.md §4.2a)."""
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "check_mfma_operands.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("v_mfma, 0 operand write(s)") == 11, r.stdout     # the eleven matrix-core kernels (Laplace: two 256-target forms, two 128-target ones; the moments kernel's seven policies) were found and are clean


def test_device_assembly_of_every_shipped_kernel_keeps_the_isa_rules():
    """tools/check_isa_rules.py over the assembly hipcc makes of EVERY translation unit of the library and of the test plugin, with the Makefile's own flags:
    (A) a kernel that issues transcendental bursts has no packed-fp32 instruction in a loop of masked exact pairs — the combination the run-to-run different
    near sums of round 3 needed (profiles/r04_near_fault_report.md); (B) no inline-asm body is the first reader of a transcendental or
    matrix-core result.  And the rule bites: the frozen reproducer of that fault (the no-fence, SLP-vectorised build of the kernel that faulted) fails it."""
    tool = os.path.join(ROOT, "tools", "check_isa_rules.py")
    r = subprocess.run([sys.executable, tool, "--shipped"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    lines = {l.split(":")[0]: l for l in r.stdout.splitlines()}
    assert len(lines) >= 17 and "yukawa_kernel.s" in lines, r.stdout                 # every unit was looked at, the plugin included
    m = re.search(r"\((\d+) with transcendental bursts", lines["centered.s"])
    assert m and int(m.group(1)) >= 2, lines["centered.s"]                           # ... and the kernels rule A is about were recognised
    bad = subprocess.run([sys.executable, tool, os.path.join(ROOT, "profiles", "r04_near_fault", "kernel_nofence.s")], capture_output=True, text=True)
    assert bad.returncode == 1 and "rule A: 4 masked-pair loop(s) with packed fp32" in bad.stdout, bad.stdout + bad.stderr
