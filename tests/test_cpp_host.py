"""The header-only C++ host surface (include/sctl_amd.hpp) driven the way the reference's src/test-fmm.cpp drives
ParticleFMM.  CPU part: the headers compile as C++11 and the program aborts loudly without a GPU (no fallback).
GPU part: its results equal the oracle on the same drand48 inputs."""
import os
import subprocess

import numpy as np
import pytest

import sctl_amd
from conftest import ROOT, rel_l2
from sctl_amd.rand48 import Rand48

SRC = os.path.join(ROOT, "tests", "cpp", "fmm_driver.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "fmm_driver")
    libdir = os.path.join(ROOT, "sctl_amd")
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + libdir, "-lsctl_amd",
           "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def _read_vector(path):
    """Vector::Write layout of the reference (vector.txx:107-119): uint64 dim, uint64 1, data."""
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:16].view(np.uint64)
    data = raw[16:].view(np.float64)
    assert hdr[1] == 1 and data.size == hdr[0]
    return data


def test_host_headers_compile_and_fail_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    if sctl_amd.device_count() > 0:
        pytest.skip("a GPU is present: the no-device abort cannot be observed")
    p = subprocess.run([exe, "50", str(tmp_path / "o")], capture_output=True, text=True)
    assert p.returncode != 0
    assert "no HIP device" in p.stderr and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_particle_fmm_driver_matches_oracle(tmp_path, O):
    exe = _build(tmp_path)
    N = 3000
    p = subprocess.run([exe, str(N), str(tmp_path / "o")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Maximum relative error: 0" in p.stdout        # Eval == EvalDirect, and Eval overwrites (fmm-wrapper.txx:85-92)
    g = Rand48(0)                                            # same draw order as the driver
    xt = g.drand48(N * 3) - 0.5
    dl_x = g.drand48(N * 3) - 0.5
    dl_n = g.drand48(N * 3) - 0.5
    dl_f = g.drand48(N * 3) - 0.5
    sl_x = g.drand48(N * 3) - 0.5
    sl_f = g.drand48(N * 3) - 0.5
    u_dl = O.eval("Stokes3D-DxU", xt, dl_x, dl_n, dl_f)
    u_sl = O.eval("Stokes3D-FxU", xt, sl_x, None, sl_f)
    # SetAccuracy(10): at least 10 digits
    assert rel_l2(_read_vector(str(tmp_path / "o_dl.bin")), u_dl) < 1e-10
    assert rel_l2(_read_vector(str(tmp_path / "o_sum.bin")), u_dl + u_sl) < 1e-10
    assert rel_l2(_read_vector(str(tmp_path / "o_acc.bin")), 2 * u_sl) < 1e-12
    M = _read_vector(str(tmp_path / "o_mat.bin")).reshape(60, 99)
    assert rel_l2(M, O.kernel_matrix("Stokes3D-DxU", xt[:99].copy(), dl_x[:60].copy(), dl_n[:60].copy())) < 1e-12
    u_h = O.eval("Helmholtz3D-FxU", xt, sl_x, None, sl_f[:2 * N].copy(), ctx=np.array([7.5, 0.3]))
    assert rel_l2(_read_vector(str(tmp_path / "o_helm.bin")), u_h) < 1e-12
    # Eval x2 + EvalDirect, the two-source Eval (2 kernels), GenericKernel::Eval x2, Helmholtz, and the 33x20 operator;
    # SCTL-convention flops = pairs * FLOPS() (generic-kernel.txx:188)
    assert "pair interactions: %d " % (N * N * 8 + 33 * 20) in p.stdout
    assert "flops: %d" % (N * N * (4 * 26 + 3 * 23 + 16) + 33 * 20 * 26) in p.stdout
