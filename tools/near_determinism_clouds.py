"""Run-to-run determinism of the tile-centred Laplace kernels through the ordinary device entry (per-call Morton sort, source splits, scatter back) on larger and
less regular problems than tools/near_determinism.py: uniform 2^20 x 2^20, clustered, points on a sphere, many sources per target; fp32 on both pipes and fp64,
single and double layer; three evaluations each, compared bit for bit."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
g = torch.Generator(device='cuda').manual_seed(3)
def clouds(kind, NT, NS):
    rng = np.random.default_rng(7)
    if kind == "uniform": return rng.random((NT, 3)), rng.random((NS, 3))
    if kind == "clustered":
        c = rng.random((50, 3)); xt = c[rng.integers(0, 50, NT)] + 1e-3 * rng.standard_normal((NT, 3))
        xs = np.concatenate([c[rng.integers(0, 50, NS // 2)] + 1e-3 * rng.standard_normal((NS // 2, 3)), 10 * rng.random((NS - NS // 2, 3))]); return xt, xs
    if kind == "surface":
        v = rng.standard_normal((NT, 3)); xt = v / np.linalg.norm(v, axis=1, keepdims=True); return xt, xt[rng.choice(NT, NS, replace=False)].copy()
total = 0
for kind, NT, NS in (("uniform", 1 << 20, 1 << 20), ("clustered", (1 << 19) + 77, 300001), ("surface", (1 << 19) + 5, 1 << 18), ("uniform", 1 << 18, (1 << 21) + 3)):
    xt, xs = clouds(kind, NT, NS)
    rng = np.random.default_rng(1)
    xn = rng.random(NS * 3) - 0.5; f = rng.random(NS) - 0.5
    for dt in (np.float32, np.float64):
        d = [torch.from_numpy(np.ascontiguousarray(a.ravel()).astype(dt)).cuda() for a in (xt, xs, xn, f)]
        bits = torch.int32 if dt == np.float32 else torch.int64
        for env in (("1", "0") if dt == np.float32 else ("1",)):
            os.environ["SCTL_AMD_MFMA_F32"] = env
            for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
                runs = [sctl_amd.eval_device(name, d[0], d[1], d[2] if name.endswith("DxU") else None, d[3]).clone() for _ in range(3)]
                nb = sum(int((r.view(bits) != runs[0].view(bits)).sum()) for r in runs[1:])
                total += nb
                print("%-10s %8d x %8d %s %-14s MFMA_F32=%s  finite %s  differing %d" % (kind, NT, NS, dt.__name__, name, env, bool(torch.isfinite(runs[0]).all()), nb), flush=True)
print("TOTAL", total)
