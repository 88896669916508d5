"""Run-to-run determinism of the evaluation kernels on the GPU: every result must be BIT-identical between repeated evaluations of the same problem (all
sums run in a fixed order; nothing is accumulated with atomics).  Written for the fault found in round 3 — near-field sums of the fp32 double-layer
kernels on the tile-centred path that differed from run to run (centered_kernel.hpp: flush_near) — which shows best when the per-wave list of pending
near sources is short, so that it is flushed between tiles all the time: build such a library with
    make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_NEARCAP=64" OUT=$PWD/tools/ab/libsctl_amd_NC64.so OBJDIR=/tmp/nc64
and run   SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_NC64.so python tools/near_determinism.py   (the shipped library without SCTL_AMD_LIB; 64 = one tile is the shortest
list the kernels allow).  -DSCTL_AMD_EXP_NO_NEAR_FENCE builds the kernels as they were before the fault was fixed; SCTL_AMD_MFMA_CB=4 runs the matrix-core
double layer with 128 targets per wave, the shape the fault was found in."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
from sctl_amd.distributed import morton_order

REPS = 4
lib = os.path.basename(os.environ.get("SCTL_AMD_LIB", "shipped library"))
g = torch.Generator(device='cuda').manual_seed(0)
bad_total = 0


def differing(fn):
    runs = [fn().clone() for _ in range(REPS)]
    assert bool(torch.isfinite(runs[0]).all()), "non-finite result"
    bits = torch.int32 if runs[0].dtype == torch.float32 else torch.int64          # bit patterns: a NaN would differ from itself as a number
    return sum(int((runs[i].view(bits) != runs[0].view(bits)).sum()) for i in range(1, REPS)), runs[0]


# (1) the tile-centred Laplace kernels: targets in Morton order handed over as a slab, so that nothing is sorted in between; both fp32 pipes
N = 1 << 18
xt = torch.rand(N * 3, dtype=torch.float64, device='cuda', generator=g); xs0 = torch.rand(N * 3, dtype=torch.float64, device='cuda', generator=g)
f0 = torch.rand(N, dtype=torch.float64, device='cuda', generator=g) - 0.5
xn0 = torch.rand(N * 3, dtype=torch.float64, device='cuda', generator=g) - 0.5
xts = xt.view(-1, 3)[morton_order(xt)].contiguous().view(-1)
for dt, digits_list in ((torch.float32, (-1, 9)), (torch.float64, (-1, 10, 3))):
    for mfma in (("1", "0") if dt == torch.float32 else ("1",)):
        os.environ["SCTL_AMD_MFMA_F32"] = mfma
        for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
            for digits in digits_list:
                for ns in (1 << 14, 70001):
                    a = [t.to(dt) for t in (xts, xs0[:3 * ns].contiguous(), xn0[:3 * ns].contiguous(), f0[:ns].contiguous())]
                    pl = sctl_amd.plan(name, 0 if dt == torch.float64 else 1, N, ns, digits, nt_whole=2 * N)
                    nb, _ = differing(lambda: sctl_amd.eval_device(name, a[0], a[1], a[2] if name.endswith("DxU") else None, a[3], digits=digits, nt_whole=2 * N))
                    bad_total += nb
                    print("%-22s %-14s %s digits %2d Ns %6d  %-12s %-22s targets differing between %d runs: %d" % (lib, name, "f32" if dt == torch.float32 else "f64", digits, ns, pl["path"],
                                                                                                                    pl["pipe"][:22], REPS, nb), flush=True)
os.environ.pop("SCTL_AMD_MFMA_F32", None)

# (2) every kernel through the exact all-pairs path, fp32 and fp64
n = 1 << 14
for name in sctl_amd.KERNEL_NAMES:
    info = sctl_amd.kernel_info(name)
    ctx = np.array([7.5, 0.3]) if info["ctx_bytes"] else None
    for dt in (torch.float32, torch.float64):
        a = [t.to(dt).contiguous() for t in (xt[:3 * n], xs0[:3 * n], xn0[:n * info["nd"]], (torch.rand(n * info["k0"], dtype=torch.float64, device='cuda', generator=g) - 0.5))]
        nb, _ = differing(lambda: sctl_amd.eval_device(name, a[0], a[1], a[2] if info["nd"] else None, a[3], ctx=ctx))
        bad_total += nb
        print("%-22s %-18s %s 2^14 x 2^14 (exact path)   values differing between %d runs: %d" % (lib, name, "f32" if dt == torch.float32 else "f64", REPS, nb), flush=True)
# (3) the batched list kernel (one launch, a wave per work item: sums in list order)
from sctl_amd.lists import grid_neighbour_lists, points_in_boxes
rng = np.random.default_rng(5)
cnt = rng.integers(1, 200, 12 ** 3)
x = points_in_boxes(12, cnt, rng)
lists = grid_neighbour_lists(12, cnt, cnt)
for name in ("Laplace3D-FxU", "Stokes3D-FxU", "Laplace3D-DxU"):
    info = sctl_amd.kernel_info(name)
    for dt in (np.float32, np.float64):
        fl = (rng.random(int(cnt.sum()) * info["k0"]) - 0.5).astype(dt)
        nl = (rng.random(int(cnt.sum()) * info["nd"]) - 0.5).astype(dt) if info["nd"] else None
        runs = [sctl_amd.eval_lists_host(name, *lists, x.astype(dt), x.astype(dt), nl, fl) for _ in range(REPS)]
        assert np.all(np.isfinite(runs[0]))
        bits = np.int32 if dt == np.float32 else np.int64
        nb = sum(int((runs[i].view(bits) != runs[0].view(bits)).sum()) for i in range(1, REPS))
        bad_total += nb
        print("%-22s %-18s %s %d lists, %d points (list kernel)   values differing between %d runs: %d" % (lib, name, "f32" if dt == np.float32 else "f64", lists[0].size, int(cnt.sum()), REPS, nb), flush=True)
print("TOTAL differing:", bad_total)
sys.exit(1 if bad_total else 0)
