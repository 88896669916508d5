"""Where does the tile-centred Laplace path start to win?  Exact kernel vs forced centred path (SCTL_AMD_CENTERED=1) by size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd
def t_eval(xt, xs, f, env, reps):
    os.environ.pop("SCTL_AMD_CENTERED", None); os.environ.update(env)
    v = torch.zeros(xt.numel() // 3, dtype=xt.dtype, device="cuda")
    sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f, v_trg=v)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for dt in (torch.float64, torch.float32):
    for nt, ns in ((1 << 14, 1 << 14), (1 << 15, 1 << 15), (1 << 16, 1 << 16), (1 << 17, 1 << 17), (1 << 18, 1 << 18), (1 << 16, 1 << 20), (1 << 17, 1 << 20), (1 << 20, 1 << 16), (1 << 20, 1 << 14)):
        g = torch.Generator(device="cuda").manual_seed(1)
        xt = torch.rand(nt * 3, dtype=dt, device="cuda", generator=g); xs = torch.rand(ns * 3, dtype=dt, device="cuda", generator=g)
        f = torch.rand(ns, dtype=dt, device="cuda", generator=g) - 0.5
        reps = max(2, int(2e11 / (nt * ns)))
        reps = min(reps, 50)
        a, b = t_eval(xt, xs, f, {"SCTL_AMD_CENTERED": "0"}, reps), t_eval(xt, xs, f, {"SCTL_AMD_CENTERED": "1"}, reps)
        print("%s Nt=2^%d Ns=2^%d: exact %.3f ms  centred %.3f ms  (%+.1f %%)" % (str(dt)[6:], nt.bit_length() - 1, ns.bit_length() - 1, a, b, 100 * (a / b - 1)), flush=True)
