"""Strong-scaling slabs: does a spatially compact target slab (Morton order) keep the tile-centred path efficient at Nt/G targets?
Times 2^20 sources against a slab of 2^20/G targets taken (a) by index from random points, (b) from the Morton-sorted order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd

def morton(x):
    q = (x.view(-1, 3).clamp(0, 1 - 1e-12) * 1024).to(torch.int64)
    key = torch.zeros(q.shape[0], dtype=torch.int64, device=x.device)
    for b in range(10):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return torch.argsort(key)

def t_eval(xt, xs, f, env):
    for k in ("SCTL_AMD_CENTERED",):
        os.environ.pop(k, None)
    os.environ.update(env)
    v = torch.zeros(xt.numel() // 3, dtype=xt.dtype, device="cuda")
    sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3

N = 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
f = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
perm = morton(xt)
xt_m = xt.view(-1, 3)[perm].contiguous().view(-1)
full = t_eval(xt, xs, f, {})
print("G=1: %.1f ms" % full, flush=True)
for G in (2, 4, 8):
    n = N // G
    a, b = xt[:3 * n].contiguous(), xt_m[:3 * n].contiguous()
    r = [t_eval(a, xs, f, {"SCTL_AMD_CENTERED": "0"}), t_eval(a, xs, f, {"SCTL_AMD_CENTERED": "1"}),
         t_eval(b, xs, f, {"SCTL_AMD_CENTERED": "0"}), t_eval(b, xs, f, {"SCTL_AMD_CENTERED": "1"})]
    print("G=%d (Nt=2^%d): index slab exact %.1f  centred %.1f | morton slab exact %.1f  centred %.1f ms   (ideal %.1f)" % (
        G, n.bit_length() - 1, *r, full / G), flush=True)
