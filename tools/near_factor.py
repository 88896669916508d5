"""Tile-centred Laplace path: time and error against the exact kernel as a function of the far/near threshold
|x_s - c|^2 > factor * Rt^2 (SCTL_AMD_EXPERIMENT_NEAR_FACTOR; the library default is 9)."""
# NOTE: the SCTL_AMD_EXPERIMENT_* switches exist only in a library built with `make -C sctl_amd/csrc EXTRA=-DSCTL_AMD_EXPERIMENTS OUT=... OBJDIR=...`
# (point SCTL_AMD_LIB at it); the shipped libsctl_amd.so ignores them.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sctl_amd
N = 1 << 20
g = torch.Generator(device="cuda").manual_seed(0)
xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g); xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
f = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
def run(env, reps=2):
    os.environ.pop("SCTL_AMD_EXPERIMENT_NEAR_FACTOR", None); os.environ.pop("SCTL_AMD_CENTERED", None)
    os.environ.update(env)
    v = sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): v = sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, v
t_ex, v_ex = run({"SCTL_AMD_CENTERED": "0"})
print("exact kernel %.1f ms" % t_ex, flush=True)
for fac in ("16", "9", "6.25", "4", "3", "2.25", "1.5"):
    t, v = run({"SCTL_AMD_EXPERIMENT_NEAR_FACTOR": fac})
    d = (v - v_ex)
    print("factor %5s: %.1f ms   rel-L2 vs exact %.2e   max rel %.2e" % (fac, t, (d.norm() / v_ex.norm()).item(), (d.abs() / v_ex.abs()).max().item()), flush=True)
