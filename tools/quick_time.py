import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import sctl_amd
def run(name, N, dt, reps=3, digits=-1):
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N*info['k0'], dtype=dt, device='cuda', generator=g)-0.5
    ctx = np.array([7.5,0.3]) if name.startswith('Helm') else None
    v = torch.zeros(N*info['k1'], dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx, digits=digits); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx, digits=digits)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    pps = N*N/(ms*1e-3); fl = sctl_amd.flops_per_pair(name)
    peak = 78.6e12 if dt == torch.float64 else 157.3e12
    print(f"{name:18s} {str(dt):14s} N={N:8d} digits={digits:3d} {ms:9.2f} ms  {pps:.3e} pairs/s  {pps*fl/1e12:6.2f} TF  {100*pps*fl/peak:5.1f}% of peak  plan={sctl_amd.plan(name, 0 if dt==torch.float64 else 1, N, N)}", flush=True)
run('Laplace3D-FxU', 1<<14, torch.float64)
run('Laplace3D-FxU', 1<<17, torch.float64)
run('Laplace3D-FxU', 1<<20, torch.float64, reps=2)
run('Laplace3D-FxU', 1<<20, torch.float64, reps=2, digits=12)
run('Laplace3D-FxU', 1<<20, torch.float32, reps=2)
for k in sctl_amd.KERNEL_NAMES[1:]:
    run(k, 1<<18, torch.float64, reps=2)
