"""List kernel with EXACTLY c points in every box (no idle target slots from ragged leaves): 27-neighbour lists of a g^3 grid, targets == sources, fp64 Laplace; packed
(SCTL_AMD_LISTS_PACK=64) against one range per wave (=0).  What is left between the two is the work-item form itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sctl_amd
from sctl_amd.lists import grid_neighbour_lists, points_in_boxes
rng = np.random.default_rng(0)
for c, g in ((8, 56), (16, 44), (32, 36), (48, 32), (64, 28), (128, 24)):
    counts = np.full(g ** 3, c)
    x = points_in_boxes(g, counts, rng)
    N = counts.sum()
    lists = grid_neighbour_lists(g, counts, counts)
    dx = torch.from_numpy(x).cuda()
    df = torch.from_numpy(rng.random(N) - 0.5).cuda()
    row = []
    for pack in ("64", "0"):
        os.environ["SCTL_AMD_LISTS_PACK"] = pack
        plan = sctl_amd.ListsPlan("Laplace3D-FxU", np.float64, *lists, N, N)
        u = torch.zeros(N, dtype=torch.float64, device="cuda")
        for _ in range(2): plan.eval_device(dx, dx, None, df, v_trg=u)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): plan.eval_device(dx, dx, None, df, v_trg=u)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        row.append((ms, 100 * plan.pairs * 11 / (ms * 1e-3) / 78.6e12, plan.work_items))
        plan.close()
    print("%4d points per box, %d^3 boxes: packed %7.3f ms %5.1f %% (%6d items)   one range per wave %7.3f ms %5.1f %% (%6d items)" % (c, g, *row[0], *row[1]), flush=True)
