"""A/B of -fno-slp-vectorize on the tile-centred unit (round 4): the shipped library against a build of the same sources WITH the SLP
vectoriser (the round-3 flags), fp32 Laplace single / double layer on both pipes and the fp64 headline, one box, alternating.
    make -C sctl_amd/csrc -j8 UNITFLAGS_centered="-mllvm -amdgpu-mfma-vgpr-form" OUT=$PWD/tools/ab/libslp.so OBJDIR=/tmp/slpbuild      (no GPU)
    python tools/ab_noslp.py                                                                                                          (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
import sctl_amd
def run(name, N, dt, reps, env=None):
    import os
    for k, v in (env or {}).items(): os.environ[k] = v
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N*info['k0'], dtype=dt, device='cuda', generator=g)-0.5
    v = torch.zeros(N*info['k1'], dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
    e1.record(); torch.cuda.synchronize()
    for k in (env or {}): del os.environ[k]
    print("%%-14s %%-8s N=2^%%d %%-22s %%9.2f ms" %% (name, str(dt)[6:], N.bit_length()-1, str(env or ''), e0.elapsed_time(e1)/reps), flush=True)
run('Laplace3D-FxU', 1<<20, torch.float64, 3)
run('Laplace3D-DxU', 1<<20, torch.float64, 2)
for name in ('Laplace3D-FxU', 'Laplace3D-DxU'):
    run(name, 1<<21, torch.float32, 3)
    run(name, 1<<21, torch.float32, 2, {'SCTL_AMD_MFMA_F32': '0'})
    run(name, 1<<21, torch.float32, 2, {'SCTL_AMD_MFMA_CB': '4'})
''' % ROOT
for rnd in range(2):
    for tag, lib in (("shipped (no SLP)", None), ("with SLP (round-3 flags)", os.path.join(ROOT, "tools", "ab", "libslp.so"))):
        env = dict(os.environ)
        if lib:
            env["SCTL_AMD_LIB"] = lib
        print("---- %s, round %d" % (tag, rnd), flush=True)
        subprocess.run([sys.executable, "-c", CODE], env=env, check=True)
