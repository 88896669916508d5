"""Helmholtz3D-FxU timing: complex and real wavenumber, fp64/fp32."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
def run(N, dt, k, reps=2):
    name = 'Helmholtz3D-FxU'
    g = torch.Generator(device='cuda').manual_seed(0)
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    f = torch.rand(N*2, dtype=dt, device='cuda', generator=g)-0.5
    ctx = np.array(k, dtype=np.float64)
    v = torch.zeros(N*2, dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, None, f, v_trg=v, ctx=ctx); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sctl_amd.eval_device(name, xt, xs, None, f, v_trg=v, ctx=ctx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    pps = N*N/(ms*1e-3); fl = sctl_amd.flops_per_pair(name)
    peak = 78.6e12 if dt == torch.float64 else 157.3e12
    print(f"{name} {str(dt):14s} k={k} N={N:8d} {ms:9.2f} ms  {pps:.3e} pairs/s  {100*pps*fl/peak:5.1f}% of peak", flush=True)
for N in (1<<18,):
    run(N, torch.float64, [7.5, 0.3]); run(N, torch.float64, [7.5, 0.0])

run(1 << 18, torch.float32, [7.5, 0.3]); run(1 << 18, torch.float32, [7.5, 0.0])
