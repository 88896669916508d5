"""What ONE rank of `bench.py --gpus G` does per step, measured on one GPU: its Morton slab of the 2^20 targets (the way
ShardedDirectSum cuts it) against all 2^20 sources through sctl_amd_eval_device_slab, for G = 1, 2, 4, 8 and the first, a middle
and the last rank.  The multi-GPU step adds one all-gather of N/G doubles per rank and an index_copy of N doubles to this."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd
from sctl_amd.distributed import morton_order, slab_bounds

N = 1 << 20
g = torch.Generator(device="cuda").manual_seed(0)
xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
f = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
perm = morton_order(xt)


def step_ms(slab, nt_whole, reps=5):
    v = torch.zeros(slab.numel() // 3, dtype=torch.float64, device="cuda")
    sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, v_trg=v, nt_whole=nt_whole)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        v.zero_()
        sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, v_trg=v, nt_whole=nt_whole)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full = step_ms(xt, N)
print("G=1: %.2f ms per step" % full, flush=True)
for G in (2, 4, 8):
    ts = []
    for r in sorted({0, G // 2, G - 1}):
        t0, t1 = slab_bounds(N, r, G)
        slab = xt.view(-1, 3)[perm[t0:t1]].contiguous().view(-1)
        ts.append(step_ms(slab, N))
    print("G=%d: rank shares %s ms; ideal %.2f; slowest rank at %.1f %% of ideal => speed-up bound %.2fx" %
          (G, " ".join("%.2f" % t for t in ts), full / G, 100 * (full / G) / max(ts), full / max(ts)), flush=True)
