"""Small problems (2^12 .. 2^16 squared, Laplace SL fp64, full precision): microseconds per step (zero + evaluate [+ reduce]) with plain stream
launches, 300 repetitions.  SCTL_AMD_LIB selects another build of the library (A/B of launch plans)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd

out = []
for logn in (12, 13, 14, 15, 16):
    N = 1 << logn
    g = torch.Generator(device="cuda").manual_seed(0)
    xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
    xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
    f = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
    v = torch.zeros(N, dtype=torch.float64, device="cuda")
    def step():
        v.zero_()
        sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    p = sctl_amd.plan("Laplace3D-FxU", 0, N, N)
    out.append("2^%d: %7.1f us (%4.1f %%, %d wg x%d)" % (logn, us, 100 * N * N * 11 / (us * 1e-6) / 78.6e12, p["workgroups"], p["src_splits"]))
print(os.path.basename(sctl_amd.library_path()), " | ".join(out), flush=True)
