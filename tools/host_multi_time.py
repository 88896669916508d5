"""Host-buffer entry over a device list (what the drop-in binding calls): wall time per call for 1, 2, 4, 8 slabs.  On a one-GPU box the slabs share
device 0, so the kernel time is the same in every row and the difference is the entry's own overhead (operator set-up, Morton order, uploads)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
import torch
n_gpu = torch.cuda.device_count()
for name, N in (("Laplace3D-FxU", 1 << 20), ("Stokes3D-FxU", 1 << 18), ("Laplace3D-FxU", 1 << 17)):
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(0)
    xt, xs, f = rng.random(N * 3), rng.random(N * 3), rng.random(N * info["k0"]) - 0.5
    for G in (1, 2, 4, 8):
        devs = [g % n_gpu for g in range(G)]
        sctl_amd.eval_host(name, xt, xs, None, f, devices=devs)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            u = sctl_amd.eval_host(name, xt, xs, None, f, devices=devs)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("%-14s N = 2^%d  %d slab(s) on %d GPU(s): %8.1f ms per call" % (name, int(np.log2(N)), G, n_gpu, ms), flush=True)
