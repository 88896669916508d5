"""A/B on ONE box: targets per lane of the vector-pipe tile-centred kernels, fp64 (the shipped library — four since late round 3, two when
profiles/r03_ab_centered_T.txt was measured — against builds with -DSCTL_AMD_EXP_CENTERED_T=2 / 3 / 6 / 8: tools/ab/libsctl_amd_T<n>.so, see
tools/ab_centered_T.sh) over kernels, sizes and accuracies: ms per evaluation, device-resident, targets unsorted (the per-call Morton sort included).
    SCTL_AMD_LIB=... python tools/ab_centered_T.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
lib = os.path.basename(os.environ.get("SCTL_AMD_LIB", "shipped"))
g = torch.Generator(device='cuda').manual_seed(0)
for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
    for lt, ls in ((18, 18), (17, 20), (20, 14), (20, 20)):
        Nt, Ns = 1 << lt, 1 << ls
        xt = torch.rand(Nt * 3, dtype=torch.float64, device='cuda', generator=g); xs = torch.rand(Ns * 3, dtype=torch.float64, device='cuda', generator=g)
        f = torch.rand(Ns, dtype=torch.float64, device='cuda', generator=g) - 0.5
        xn = (torch.rand(Ns * 3, dtype=torch.float64, device='cuda', generator=g) - 0.5) if name.endswith("DxU") else None
        out = []
        for digits in (-1, 10):
            v = torch.zeros(Nt, dtype=torch.float64, device='cuda')
            sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, digits=digits); torch.cuda.synchronize()
            reps = 3 if lt + ls >= 38 else 10
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, digits=digits)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
        print("%-8s %-14s 2^%d x 2^%d  full precision %9.3f ms   10 digits %9.3f ms   path %s" % (lib, name, lt, ls, out[0], out[1], sctl_amd.plan(name, 0, Nt, Ns)["path"]), flush=True)
