"""Near-field operator application (sctl_amd_near_apply_device): achieved HBM rate = bytes of K_near / kernel time.
usage: time_near.py [Nelem nodes_per_elem near_targets_per_elem k0 k1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sctl_amd

def run(nelem, nds, near, k0, k1, dtype=np.float64, reps=10):
    rng = np.random.default_rng(0)
    nds_a = np.full(nelem, nds, dtype=np.int64); near_a = np.full(nelem, near, dtype=np.int64)
    n_near = nelem * near
    ntrg = max(1, n_near // 8)                                  # every target is near ~8 elements
    K = rng.standard_normal(nelem * nds * k0 * near * k1).astype(dtype)
    trg = rng.integers(0, ntrg, n_near)
    order = np.argsort(trg, kind="stable"); cnt = np.bincount(trg, minlength=ntrg); dsp = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    op = sctl_amd.NearOp(k0, k1, nds_a, near_a, K, order, cnt, dsp)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    F = torch.randn(op.density_len, dtype=tdt, device="cuda"); U = torch.zeros(op.potential_len, dtype=tdt, device="cuda")
    op.apply_device(F, U); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): op.apply_device(F, U)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("Nelem %6d  block %4d x %5d  %s  K_near %7.1f MB  workgroups %7d  %8.3f ms  %7.1f GB/s  (%.1f %% of 8 TB/s)" % (
        nelem, nds * k0, near * k1, np.dtype(dtype).name, op.operator_bytes / 1e6, op.workgroups, ms, op.operator_bytes / ms / 1e6, op.operator_bytes / ms / 1e6 / 80), flush=True)
    op.close()

if len(sys.argv) > 5:
    run(*[int(a) for a in sys.argv[1:6]])
else:
    run(2048, 48, 400, 3, 3)          # Stokes-like: 144 x 1200 blocks, 2.8 GB
    run(8192, 24, 200, 1, 1)          # Laplace-like: 24 x 200 blocks, 0.3 GB
    run(2048, 48, 400, 3, 3, np.float32)
    run(20000, 16, 30, 1, 1)          # many small blocks
