"""PCIe-inclusive rate of the host-buffer entry sctl_amd_eval_host (the GenericKernel::Eval drop-in) on the headline
workload — reported in DESIGN.md next to the device-resident number; never the bench `value`."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import sctl_amd  # noqa: E402

for logn in (14, 17, 20):
    N = 1 << logn
    rng = np.random.default_rng(0)
    xt, xs, f = rng.random(N * 3), rng.random(N * 3), rng.random(N) - 0.5
    sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f)          # warm-up (HIP context, library load)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        sctl_amd.eval_host("Laplace3D-FxU", xt, xs, None, f)
        best = min(best, time.perf_counter() - t)
    print("sctl_amd_eval_host Laplace3D-FxU f64 N=2^%d: %.3f ms wall (upload + kernel + download + accumulate) = %.3e pair-interactions/s" %
          (logn, best * 1e3, N * N / best), flush=True)
