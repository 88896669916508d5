"""Time one kernel at one size on the GPU: python tools/time_one.py <kernel> <log2 N | log2Nt,log2Ns> <f64|f32> [digits]
(SCTL_AMD_LIB selects another build of the library, SCTL_AMD_CENTERED=0 disables the tile-centred Laplace path)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd  # noqa: E402

name, logs, dts = sys.argv[1], sys.argv[2], sys.argv[3]
logn = int(logs.split(',')[0])
logns = int(logs.split(',')[-1])
digits = int(sys.argv[4]) if len(sys.argv) > 4 else -1
dt = torch.float64 if dts == 'f64' else torch.float32
N, Ns = 1 << logn, 1 << logns
info = sctl_amd.kernel_info(name)
g = torch.Generator(device='cuda').manual_seed(0)
xt = torch.rand(N * 3, dtype=dt, device='cuda', generator=g)
xs = torch.rand(Ns * 3, dtype=dt, device='cuda', generator=g)
xn = torch.rand(Ns * info['nd'], dtype=dt, device='cuda', generator=g) - 0.5
f = torch.rand(Ns * info['k0'], dtype=dt, device='cuda', generator=g) - 0.5
ctx = np.array([7.5, 0.3]) if name.startswith('Helm') else None
v = torch.zeros(N * info['k1'], dtype=dt, device='cuda')
sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx, digits=digits)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps):
    sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v, ctx=ctx, digits=digits)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
pps = N * Ns / (ms * 1e-3)
fl = sctl_amd.flops_per_pair(name)
peak = 78.6e12 if dts == 'f64' else 157.3e12
print(f"{sctl_amd.library_path().split('/')[-1]:36s} {name:18s} {dts} Nt=2^{logn} Ns=2^{logns} digits={digits:3d} {ms:9.2f} ms  {pps:.3e} pairs/s  {100 * pps * fl / peak:5.1f}% of peak", flush=True)
