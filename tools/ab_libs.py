"""A/B of scratch builds of the library on one box: every tools/ab/libs/lib*.so (built with `make EXTRA=... OUT=... OBJDIR=...`) against the shipped one, one child process
each (SCTL_AMD_LIB), one kernel, fp64, 2^18 and 2^20 points, full precision.   usage: ab_libs.py <kernel name> [centred 0|1]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch, sctl_amd
name = sys.argv[1]
info = sctl_amd.kernel_info(name)
for N, reps in ((1 << 18, 6), (1 << 20, 2)):
    g = torch.Generator(device='cuda').manual_seed(0)
    dt = torch.float64
    xt = torch.rand(N*3, dtype=dt, device='cuda', generator=g); xs = torch.rand(N*3, dtype=dt, device='cuda', generator=g)
    xn = torch.rand(N*info['nd'], dtype=dt, device='cuda', generator=g)-0.5; f = torch.rand(N*info['k0'], dtype=dt, device='cuda', generator=g)-0.5
    v = torch.zeros(N*info['k1'], dtype=dt, device='cuda')
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1)/reps)
    pl = sctl_amd.plan(name, 0, N, N)
    print("   2^%%d: %%9.2f ms (%%s, T=%%d, checksum %%.12e)" %% (N.bit_length()-1, best, pl['path'], pl['trg_per_lane'], float(v.double().abs().sum())), end='')
print()
''' % ROOT
name = sys.argv[1]
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "ab", "libs", "lib*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib: env["SCTL_AMD_LIB"] = lib
    for c in ((sys.argv[2],) if len(sys.argv) > 2 else (("1", "0") if lib is None else ("1",))):
        env["SCTL_AMD_CENTERED"] = c
        sys.stdout.write("%-14s %-7s" % (os.path.basename(lib) if lib else "shipped", "centred" if c == "1" else "exact")); sys.stdout.flush()
        subprocess.run([sys.executable, "-c", CHILD, name], env=env)
