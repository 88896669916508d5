import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sctl_amd
from sctl_amd.distributed import morton_order
lib = os.path.basename(os.environ.get("SCTL_AMD_LIB", "shipped(T=4)"))
g = torch.Generator(device='cuda').manual_seed(0)
N = 1 << 20
xt = torch.rand(N * 3, dtype=torch.float64, device='cuda', generator=g); xs = torch.rand(N * 3, dtype=torch.float64, device='cuda', generator=g)
f = torch.rand(N, dtype=torch.float64, device='cuda', generator=g) - 0.5
xts = xt.view(-1, 3)[morton_order(xt)].contiguous()
for G in (2, 4, 8):
    for digits in (-1, 10):
        ts = []
        for r in (0, G // 2, G - 1):
            t0, t1 = N * r // G, N * (r + 1) // G
            slab = xts[t0:t1].contiguous().view(-1)
            v = torch.zeros(t1 - t0, dtype=torch.float64, device='cuda')
            sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, v_trg=v, digits=digits, nt_whole=N); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): sctl_amd.eval_device("Laplace3D-FxU", slab, xs, None, f, v_trg=v, digits=digits, nt_whole=N)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        print("%-22s G=%d digits %2d: rank shares (first, middle, last) %s ms" % (lib, G, digits, " ".join("%.2f" % t for t in ts)), flush=True)
