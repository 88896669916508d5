"""All ten kernels in fp32 at 2^18 x 2^18 (the exact all-pairs kernel, or the tile-centred / matrix-core path where a kernel has one): ms and % of the 157.3 TF fp32 vector peak by the flop
convention of SURVEY.md §8d.  The reference runs every functor at Real = float as well (generic-kernel.txx:76)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sctl_amd
N = 1 << 18
for name in sctl_amd.KERNEL_NAMES:
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    dt = torch.float32
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N*info['k0'], dtype=dt, device='cuda', generator=g)-0.5
    ctx = np.array([7.5, 0.3]) if name.startswith('Helm') else None
    v = torch.zeros(N*info['k1'], dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    fl = sctl_amd.flops_per_pair(name)
    pl = sctl_amd.plan(name, 1, N, N)
    print("%-18s fp32 2^18: %8.2f ms  %5.1f %% of 157.3 TF  (%s, %s, T=%d, %d splits)" % (name, ms, 100*N*N*fl/(ms*1e-3)/157.3e12, pl['path'], 'matrix cores' if pl['pipe'].startswith('bf16') else 'vector pipe', pl['trg_per_lane'], pl['src_splits']), flush=True)
