import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, sctl_amd
def run(name, ln, cb):
    N = 1 << ln
    if cb: os.environ['SCTL_AMD_MFMA_CB'] = cb
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    dt = torch.float32
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N, dtype=dt, device='cuda', generator=g)-0.5
    v = torch.zeros(N, dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 10 if ln <= 19 else 4
    e0.record()
    for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
    e1.record(); torch.cuda.synchronize()
    os.environ.pop('SCTL_AMD_MFMA_CB', None)
    return e0.elapsed_time(e1)/reps
for name in ('Laplace3D-FxU', 'Laplace3D-DxU'):
    for ln in (18, 19, 20, 21):
        a, b = run(name, ln, None), run(name, ln, '4')
        a2, b2 = run(name, ln, None), run(name, ln, '4')
        fl = sctl_amd.flops_per_pair(name)
        print("%-14s fp32 2^%d: 256 targets per wave %8.2f / %8.2f ms (%4.1f %%)   128 per wave %8.2f / %8.2f ms (%4.1f %%)" % (name, ln, a, a2, 100*(1<<ln)**2*fl/(min(a,a2)*1e-3)/157.3e12, b, b2, 100*(1<<ln)**2*fl/(min(b,b2)*1e-3)/157.3e12), flush=True)
