"""Batched list evaluation across kernels and leaf sizes: 2^21 uniform points in g^3 boxes (g = 16, 32, 64: ~512, ~64, ~8 points per
box), every box against itself and its neighbours, targets == sources, fp64 full precision.  Prints ms, pairs/s and the fraction of the
fp64 vector peak by the flop convention of SURVEY.md §8d."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sctl_amd
from sctl_amd.lists import grid_neighbour_lists

N = 1 << 21
rng = np.random.default_rng(0)
GRIDS = [int(g) for g in os.environ.get("LISTS_GRIDS", "16,32,64").split(",")]
KERNELS = os.environ.get("LISTS_KERNELS", "Laplace3D-FxU,Stokes3D-FxU,Stokes3D-DxU").split(",")
for grid in GRIDS:
    x = rng.random((N, 3))
    box = (np.floor(x[:, 0] * grid) * grid + np.floor(x[:, 1] * grid)) * grid + np.floor(x[:, 2] * grid)
    order = np.argsort(box, kind="stable")
    x = x[order].ravel().copy()
    counts = np.bincount(box.astype(np.int64), minlength=grid ** 3)
    lists = grid_neighbour_lists(grid, counts, counts)
    dx = torch.from_numpy(x).cuda()
    for name in KERNELS:
        info = sctl_amd.kernel_info(name)
        plan = sctl_amd.ListsPlan(name, np.float64, *lists, N, N)
        dn = torch.from_numpy(rng.random(N * info["nd"]) - 0.5).cuda()
        df = torch.from_numpy(rng.random(N * info["k0"]) - 0.5).cuda()
        u = torch.zeros(N * info["k1"], dtype=torch.float64, device="cuda")
        for _ in range(2):
            plan.eval_device(dx, dx, dn, df, v_trg=u)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            plan.eval_device(dx, dx, dn, df, v_trg=u)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        fpp = sctl_amd.flops_per_pair(name)
        print("%-14s %3d^3 boxes (~%4d pts)  %7d lists %8d items  %8.2f ms  %.3e pairs/s  %5.1f %% of peak" %
              (name, grid, N // grid ** 3, lists[0].size, plan.work_items, ms, plan.pairs / ms * 1e3, 100 * plan.pairs * fpp / (ms * 1e-3) / 78.6e12), flush=True)
        plan.close()
