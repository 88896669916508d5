"""fp32 Stokeslet family and stresslet: tile-centred matrix-core path (r2 and the dot products as split-bf16 contractions, four far moments on the vector pipe) against the exact fp32 kernel
(SCTL_AMD_CENTERED=0) and the fp64 exact kernel, one box.  Prints ms, % of the 157.3 TF fp32 vector peak by the flop convention, rel-L2 against fp64, and whether six
repeated evaluations are bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sctl_amd
def run(name, N, reps):
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    xt64 = torch.rand(N*3, dtype=torch.float64, device='cuda', generator=g); xs64 = torch.rand(N*3, dtype=torch.float64, device='cuda', generator=g)
    f64 = torch.rand(N*info['k0'], dtype=torch.float64, device='cuda', generator=g)-0.5
    xn = (torch.rand(N*info['nd'], dtype=torch.float64, device='cuda', generator=g)-0.5).float() if info['nd'] else None
    xt, xs, f = xt64.float(), xs64.float(), f64.float()
    ref = sctl_amd.eval_device(name, xt.double(), xs.double(), None if xn is None else xn.double(), f.double())
    out = {}
    for mode in ('1', '0'):
        os.environ['SCTL_AMD_CENTERED'] = mode
        v = torch.zeros(N*info['k1'], dtype=torch.float32, device='cuda')
        sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
        first = v.clone()
        same = all(bool((sctl_amd.eval_device(name, xt, xs, xn, f).view(torch.int32) == first.view(torch.int32)).all()) for _ in range(5))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
        e1.record(); torch.cuda.synchronize()
        out[mode] = (e0.elapsed_time(e1)/reps, float((first.double()-ref).norm()/ref.norm()), sctl_amd.plan(name, 1, N, N), same, bool(torch.isfinite(first).all()))
    del os.environ['SCTL_AMD_CENTERED']
    fl = sctl_amd.flops_per_pair(name)
    a, b = out['1'], out['0']
    print("%-14s fp32 N=2^%d  %s/%s T=%d %d splits %9.2f ms (%5.1f %%) err %.1e finite %s repeat-identical %s | exact %9.2f ms (%5.1f %%) err %.1e  -> x%.2f" % (
        name, N.bit_length()-1, a[2]['path'], a[2]['pipe'][:12], a[2]['trg_per_lane'], a[2]['src_splits'], a[0], 100*N*N*fl/(a[0]*1e-3)/157.3e12, a[1], a[4], a[3],
        b[0], 100*N*N*fl/(b[0]*1e-3)/157.3e12, b[1], b[0]/a[0]), flush=True)
for name in (sys.argv[1:] or ['Stokes3D-FxU', 'Stokes3D-FSxU', 'Stokes3D-FxUP', 'Stokes3D-DxU', 'Stokes3D-FxT', 'Laplace3D-FxdU', 'Laplace3D-FDxUdU']):
    run(name, 1 << 18, 5); run(name, 1 << 20, 2)
