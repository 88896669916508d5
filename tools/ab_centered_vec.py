"""Tile-centred against exact path (SCTL_AMD_CENTERED=0) for kernels with vector outputs (round 4), fp64, one box.
usage: ab_centered_vec.py [kernel names ...]   (default: the Laplace gradient; SCTL_AMD_LIB selects a scratch build of the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sctl_amd
def run(name, N, reps, digits=-1):
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    dt = torch.float64
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N*info['k0'], dtype=dt, device='cuda', generator=g)-0.5
    out = {}
    for mode in ('1', '0'):
        os.environ['SCTL_AMD_CENTERED'] = mode
        v = torch.zeros(N*info['k1'], dtype=dt, device='cuda')
        sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, digits=digits); torch.cuda.synchronize()
        first = v.clone()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, digits=digits)
        e1.record(); torch.cuda.synchronize()
        out[mode] = (e0.elapsed_time(e1)/reps, first, sctl_amd.plan(name, 0, N, N, digits=digits))
    del os.environ['SCTL_AMD_CENTERED']
    fl = sctl_amd.flops_per_pair(name)
    d = float((out['1'][1]-out['0'][1]).norm()/out['0'][1].norm())
    print("%-18s N=2^%d digits %3d  centred %9.2f ms (%5.1f %%, T=%d, %d splits)   exact %9.2f ms (%5.1f %%)   rel-L2 between them %.1e" % (name, N.bit_length()-1, digits,
          out['1'][0], 100*N*N*fl/(out['1'][0]*1e-3)/78.6e12, out['1'][2]['trg_per_lane'], out['1'][2]['src_splits'], out['0'][0], 100*N*N*fl/(out['0'][0]*1e-3)/78.6e12, d), flush=True)
for name in (sys.argv[1:] or ['Laplace3D-FxdU']):
    run(name, 1 << 18, 5); run(name, 1 << 18, 5, 10); run(name, 1 << 20, 2); run(name, 1 << 20, 2, 10)
