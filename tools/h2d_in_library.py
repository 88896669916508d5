"""Does an H2D copy straight from the caller's (pageable, reused, rewritten in place) arrays ever deliver stale contents INSIDE the
library's host entries?  Round 1 saw that 3-4 times per 3000 transfers and introduced pinned staging; the stand-alone stress
(tools/ubench/h2d_reuse.hip) does not show it.  This is the test loop of tests/test_gpu_parity.py::test_reused_host_buffers_are_always_re_read,
longer, counting instead of asserting; run it once with the shipped library and once with a build whose upload() skips the staging:
    make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_NO_STAGING" OUT=$PWD/tools/ab/libsctl_amd_nostage.so OBJDIR=/tmp/nostage
    SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_nostage.so python tools/h2d_in_library.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import sctl_amd

O = oracle.restatement()
name, N, iters = "Stokes3D-DxU", 3000, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
info = sctl_amd.kernel_info(name)
rng = np.random.default_rng(23)
xt, xa, xb = rng.random(N * 3), rng.random(N * 3), rng.random(N * 3)
na, fa, fb = rng.random(N * 3) - 0.5, rng.random(N * 3) - 0.5, rng.random(N * 3) - 0.5
ref = {(u, v): O.eval(name, xt, xb if u else xa, na, fb if v else fa) for u in (0, 1) for v in (0, 1)}
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
op = sctl_amd.DirectOp(name)
op.set_targets(xt)
buf, fbuf = xa.copy(), fa.copy()
bad_op = bad_host = 0
for it in range(iters):
    ub, uf = it & 1, (it >> 1) & 1
    buf[:] = xb if ub else xa
    fbuf[:] = fb if uf else fa
    op.set_sources(buf, na)
    bad_op += rel(op.eval(fbuf), ref[(ub, uf)]) > 1e-12
    bad_host += rel(sctl_amd.eval_host(name, xt, buf, na, fbuf), ref[(ub, uf)]) > 1e-12
print("%s: %d iterations, wrong results: operator handle %d, one-shot host entry %d" % (os.environ.get("SCTL_AMD_LIB", "shipped library"), iters, bad_op, bad_host))
