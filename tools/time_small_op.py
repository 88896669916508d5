"""cfg 1 (2^14 x 2^14 Laplace SL fp64): where do the microseconds of one benchmark step go?  The same evaluation through
(a) ShardedDirectSum.eval_slab as bench.py calls it, (b) sctl_amd.eval_device, (c) the bare ctypes entry with everything precomputed;
each with and without the zero fill, plus the host-side cost of issuing a step without waiting for the device."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd
from sctl_amd import api
from sctl_amd.distributed import ShardedDirectSum

N = 1 << 14
g = torch.Generator(device="cuda").manual_seed(0)
xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
xn = torch.empty(0, dtype=torch.float64, device="cuda")
f = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
v = torch.zeros(N, dtype=torch.float64, device="cuda")
op = ShardedDirectSum("Laplace3D-FxU")
op.set_targets(xt)
lib = api.lib()
kid = sctl_amd.kernel_info("Laplace3D-FxU")["id"]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = [C.c_void_p(t.data_ptr()) for t in (xt, xs, f, v)]


def a_step():
    op.eval_slab(xt, xs, xn, f, v)


def b_step():
    v.zero_()
    sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v)


def c_step():
    v.zero_()
    lib.sctl_amd_eval_device_slab(kid, 0, N, N, N, ptr[0], ptr[1], None, ptr[2], ptr[3], -1, None, 0, st)


def c_nozero():
    lib.sctl_amd_eval_device_slab(kid, 0, N, N, N, ptr[0], ptr[1], None, ptr[2], ptr[3], -1, None, 0, st)


for name, fn in (("eval_slab (bench.py's step)", a_step), ("eval_device", b_step), ("bare ABI call + zero", c_step), ("bare ABI call, no zero", c_nozero)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%-30s %7.1f us per step on the device, %6.1f us of host time to issue it" % (name, t_all / reps * 1e6, t_issue / reps * 1e6), flush=True)
