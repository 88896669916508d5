"""A/B on ONE box: fp32 Laplace single (and, with --dl, double) layer through the tile-centred path, the far pairs' contractions on the bf16 matrix
cores (centered_mfma_kernel.hpp, the shipped default) against the packed-VALU kernel (SCTL_AMD_MFMA_F32=0), with each result's rel-L2 distance to the
fp64 kernel on the same fp32-rounded inputs.
    python tools/ab_mfma_f32.py [--dl] [--mfma-only] [log2 N ...]      (default 20 21)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd

DL = "--dl" in sys.argv
NAME = "Laplace3D-DxU" if DL else "Laplace3D-FxU"
FPP = sctl_amd.flops_per_pair(NAME)

def run(N, reps):
    g = torch.Generator(device='cuda').manual_seed(0)
    xt = torch.rand(N * 3, dtype=torch.float32, device='cuda', generator=g); xs = torch.rand(N * 3, dtype=torch.float32, device='cuda', generator=g)
    f = torch.rand(N, dtype=torch.float32, device='cuda', generator=g) - 0.5
    xn = (torch.rand(N * 3, dtype=torch.float32, device='cuda', generator=g) - 0.5) if DL else None
    sel = torch.arange(0, N, max(1, N // 4096), device='cuda')[:4096]
    ref = sctl_amd.eval_device(NAME, xt.double().view(-1, 3)[sel].contiguous().view(-1), xs.double(), xn.double() if DL else None, f.double())
    for tag, env in ((("bf16 MFMA", None), ("bf16 MFMA", None)) if "--mfma-only" in sys.argv else (("bf16 MFMA", None), ("packed VALU", "0"), ("bf16 MFMA", None), ("packed VALU", "0"))):
        if env is None: os.environ.pop("SCTL_AMD_MFMA_F32", None)
        else: os.environ["SCTL_AMD_MFMA_F32"] = env
        v = torch.zeros(N, dtype=torch.float32, device='cuda')
        sctl_amd.eval_device(NAME, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
        err = float((v[sel].double() - ref).norm() / ref.norm())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): sctl_amd.eval_device(NAME, xt, xs, xn, f, v_trg=v)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%s 2^%d x 2^%d fp32  %-14s %9.2f ms  %.3e pairs/s  %5.1f %% of 157.3 TF   rel-L2 vs fp64 kernel %.2e" % (NAME, N.bit_length() - 1, N.bit_length() - 1, tag, ms, N * N / (ms * 1e-3), N * N / (ms * 1e-3) * FPP / 157.3e12 * 100, err), flush=True)
    os.environ.pop("SCTL_AMD_MFMA_F32", None)

for lg in ([int(a) for a in sys.argv[1:] if not a.startswith('--')] or [20, 21]):
    run(1 << lg, 3 if lg <= 21 else 1)
