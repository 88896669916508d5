"""Device-resident operator (sctl_amd_op_*) with the targets kept in Morton order: exact kernel vs tile-centred kernel without the
per-call sort, by size (host entry: includes the density upload and the potential download)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sctl_amd
rng = np.random.default_rng(0)
for name in ("Laplace3D-FxU", "Laplace3D-DxU"):
    for nt, ns in ((1 << 14, 1 << 14), (1 << 15, 1 << 15), (1 << 16, 1 << 16), (1 << 17, 1 << 17), (1 << 18, 1 << 18), (1 << 15, 1 << 18), (1 << 17, 1 << 14)):
        xt, xs, f = rng.random(nt * 3), rng.random(ns * 3), rng.random(ns) - 0.5
        xn = rng.random(ns * 3) - 0.5 if name.endswith("DxU") else None
        res = {}
        for tag, env in (("exact", "0"), ("centred", "1")):
            os.environ["SCTL_AMD_CENTERED"] = env
            op = sctl_amd.DirectOp(name)
            op.set_targets(xt); op.set_sources(xs, xn)
            u = op.eval(f)
            for _ in range(3): op.eval(f, u)
            reps = max(3, min(40, int(3e10 / (nt * ns))))
            best = 1e30
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(reps): op.eval(f, u)
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            res[tag] = (best, u.copy())
            op.close()
        os.environ.pop("SCTL_AMD_CENTERED")
        d = np.linalg.norm(res["exact"][1] - res["centred"][1]) / np.linalg.norm(res["exact"][1])
        print("%s Nt=2^%d Ns=2^%d: exact %.3f ms  centred(presorted) %.3f ms  (%+.1f %%)  rel-L2 %.1e" % (name, nt.bit_length() - 1, ns.bit_length() - 1, res["exact"][0], res["centred"][0], 100 * (res["exact"][0] / res["centred"][0] - 1), d), flush=True)
