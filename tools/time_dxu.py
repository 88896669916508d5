import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sctl_amd
def t_eval(name, xt, xs, xn, f, env, reps):
    os.environ.pop("SCTL_AMD_CENTERED", None); os.environ.update(env)
    v = torch.zeros(xt.numel() // 3, dtype=xt.dtype, device="cuda")
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, v
for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
    for dt in (torch.float64, torch.float32):
        for n in (1 << 18, 1 << 20):
            g = torch.Generator(device="cuda").manual_seed(1)
            xt = torch.rand(n * 3, dtype=dt, device="cuda", generator=g); xs = torch.rand(n * 3, dtype=dt, device="cuda", generator=g)
            xn = torch.rand(n * 3, dtype=dt, device="cuda", generator=g) - 0.5 if name.endswith("DxU") else None
            f = torch.rand(n, dtype=dt, device="cuda", generator=g) - 0.5
            reps = 4 if n < (1 << 20) else 2
            (a, va), (b, vb) = t_eval(name, xt, xs, xn, f, {"SCTL_AMD_CENTERED": "0"}, reps), t_eval(name, xt, xs, xn, f, {"SCTL_AMD_CENTERED": "1"}, reps)
            print("%s %s N=2^%d: exact %.2f ms  centred %.2f ms (%+.1f %%)  rel-L2 %.2e" % (name, str(dt)[6:], n.bit_length() - 1, a, b, 100 * (a / b - 1), ((va - vb).norm() / va.norm()).item()), flush=True)
