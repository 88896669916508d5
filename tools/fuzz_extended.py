"""One-off extended randomised parity run (not part of the test suite: ~5 minutes of GPU + host oracle time): list evaluation with random
kernels / precisions / digits / ragged lists (tiny, mid and large target ranges, so every item kind of lists_kernel.hpp runs), the all-pairs
entries at random sizes, every tile-centred form forced onto small ragged problems, and the fused far + near potential on random operators over 1-3 slabs.  Every result against the CPU oracle.
    python tools/fuzz_extended.py [cases per family, default 150]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
import sctl_amd

O = oracle.restatement()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))
worst = {}


def note(family, err, tol, what):
    worst[family] = max(worst.get(family, 0.0), err / tol)
    if not (err <= tol):
        print("FAIL", family, what, "err %.3e tol %.1e" % (err, tol), flush=True)
        sys.exit(1)


def ctx_of(name, rng):
    return np.array([rng.uniform(-20, 20), rng.uniform(0, 3)]) if name.startswith("Helmholtz") else None


t0 = time.time()
for c in range(n_cases):                                   # ---- lists
    rng = np.random.default_rng(50000 + c)
    name = sctl_amd.KERNEL_NAMES[int(rng.integers(0, len(sctl_amd.KERNEL_NAMES)))]
    info = sctl_amd.kernel_info(name)
    dt = np.float64 if rng.random() < 0.65 else np.float32
    digits = int(rng.choice([-1, -1, 12, 9, 5]))
    Ns = int(rng.integers(1, 2500))
    nbox = int(rng.integers(1, 40))
    kind = rng.integers(0, 3, nbox)                        # tiny (replicas), mid (one per lane), large (two per lane) target ranges
    tlen = np.where(kind == 0, rng.integers(0, 33, nbox), np.where(kind == 1, rng.integers(33, 65, nbox), rng.integers(65, 400, nbox)))
    tstart = np.cumsum(np.concatenate([[0], tlen[:-1]]) + rng.integers(0, 3, nbox))
    Nt = int(tstart[-1] + tlen[-1] + 1)
    to, tc, so, sc = [], [], [], []
    self_lists = rng.random() < 0.5
    for b in range(nbox):
        for _ in range(int(rng.integers(0, 30))):
            n = min(Ns, int(rng.integers(0, 4)) if rng.random() < 0.4 else int(rng.integers(0, min(200, Ns) + 1)))
            s0 = int(rng.integers(0, Ns - n + 1))
            to.append(tstart[b]); tc.append(tlen[b]); so.append(s0); sc.append(n)
    lists = [np.array(a, dtype=np.int64) for a in (to, tc, so, sc)]
    xs = rng.random(Ns * 3).astype(dt)
    xt = rng.random(Nt * 3).astype(dt)
    if self_lists:                                          # coincident points: targets copied from sources
        k = min(Nt, Ns)
        xt[:k * 3] = xs[:k * 3]
    xn, f = (rng.random(Ns * info["nd"]) - 0.5).astype(dt), (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
    ctx = ctx_of(name, rng)
    u = sctl_amd.eval_lists_host(name, *lists, xt, xs, xn, f, digits=digits, ctx=ctx)
    ref = np.zeros(Nt * info["k1"])
    x64 = [a.astype(np.float64) for a in (xt, xs, xn, f)]
    k0, k1, nd = info["k0"], info["k1"], info["nd"]
    for a, n, s, m in zip(*lists):
        if n and m:
            O.eval(name, x64[0][a * 3:(a + n) * 3].copy(), x64[1][s * 3:(s + m) * 3].copy(), x64[2][s * nd:(s + m) * nd].copy(), x64[3][s * k0:(s + m) * k0].copy(),
                   v_trg=ref[a * k1:(a + n) * k1], ctx=ctx, nthreads=1)
    tol = (1e-12 if digits < 0 or digits >= 15 else 10.0 * 10.0 ** -digits) if dt == np.float64 else (1e-4 if digits == 5 else 3e-5)
    if np.linalg.norm(ref) > 0:
        note("lists", rel(u, ref), tol, (c, name, dt.__name__, digits, Nt, Ns, len(to)))
print("lists: %d cases, worst err/tol %.3f  (%.0f s)" % (n_cases, worst.get("lists", 0), time.time() - t0), flush=True)

t0 = time.time()
for c in range(n_cases):                                   # ---- all pairs, host / operator entries
    rng = np.random.default_rng(60000 + c)
    name = sctl_amd.KERNEL_NAMES[int(rng.integers(0, len(sctl_amd.KERNEL_NAMES)))]
    info = sctl_amd.kernel_info(name)
    dt = np.float64 if rng.random() < 0.65 else np.float32
    digits = int(rng.choice([-1, -1, 12, 9, 5]))
    Nt, Ns = int(rng.integers(1, 20000)), int(rng.integers(1, 9000))
    xt, xs = rng.random(Nt * 3).astype(dt), rng.random(Ns * 3).astype(dt)
    xn, f = (rng.random(Ns * info["nd"]) - 0.5).astype(dt), (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
    ctx = ctx_of(name, rng)
    if rng.random() < 0.5:
        u = sctl_amd.eval_host(name, xt, xs, xn, f, digits=digits, ctx=ctx, devices=[0] * int(rng.integers(1, 4)))
    else:
        op = sctl_amd.DirectOp(name, dt, devices=[0] * int(rng.integers(1, 4)), ctx=ctx)
        op.set_targets(xt); op.set_sources(xs, xn)
        u = op.eval(f, digits=digits)
        op.close()
    sel = rng.choice(Nt, min(Nt, 400), replace=False)
    ref = O.eval(name, xt.reshape(-1, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), xn.astype(np.float64), f.astype(np.float64), ctx=ctx).reshape(sel.size, -1)
    tol = (1e-12 if digits < 0 or digits >= 15 else 10.0 * 10.0 ** -digits) if dt == np.float64 else (1e-4 if digits == 5 else 3e-5)
    note("all-pairs", rel(u.reshape(Nt, -1)[sel], ref), tol, (c, name, dt.__name__, digits, Nt, Ns))
print("all-pairs: %d cases, worst err/tol %.3f  (%.0f s)" % (n_cases, worst.get("all-pairs", 0), time.time() - t0), flush=True)

t0 = time.time()
FORMS = [("Laplace3D-FxU", np.float64), ("Laplace3D-FxU", np.float32), ("Laplace3D-DxU", np.float64), ("Laplace3D-DxU", np.float32), ("Laplace3D-FxdU", np.float64),
         ("Stokes3D-FxUP", np.float64), ("Stokes3D-FxU", np.float32), ("Stokes3D-FSxU", np.float32), ("Stokes3D-FxUP", np.float32), ("Stokes3D-DxU", np.float32), ("Stokes3D-FxT", np.float32), ("Laplace3D-FxdU", np.float32), ("Laplace3D-FDxUdU", np.float32)]
os.environ["SCTL_AMD_CENTERED"] = "1"                      # ---- every tile-centred form forced onto small ragged problems (>= 128 targets, >= 64 sources)
for c in range(n_cases):
    rng = np.random.default_rng(80000 + c)
    name, dt = FORMS[int(rng.integers(0, len(FORMS)))]
    info = sctl_amd.kernel_info(name)
    f64 = dt == np.float64
    digits = int(rng.choice([-1, -1, 12, 9, 5])) if (f64 or name in ("Laplace3D-FxU", "Laplace3D-DxU")) else int(rng.choice([-1, -1, 5]))
    Nt, Ns = int(rng.integers(128, 6000)), int(rng.integers(64, 6000))
    scale = float(rng.choice([1.0, 1e-2, 1e2]))
    xs = (scale * rng.random(Ns * 3)).astype(dt)
    kind = int(rng.integers(0, 3))
    xt = (scale * (rng.random(Nt * 3) if kind == 0 else 0.1 * rng.random(Nt * 3) + 0.45 if kind == 1 else np.repeat(rng.random((max(1, Nt // 50), 3)), 50, axis=0)[:Nt].ravel() if Nt >= 50 else rng.random(Nt * 3))).astype(dt)
    if xt.size != Nt * 3:
        xt = (scale * rng.random(Nt * 3)).astype(dt)
    if rng.random() < 0.4:
        k = min(Nt, Ns, int(rng.integers(1, 300)))
        xt[:k * 3] = xs[:k * 3]
    xn, f = (rng.random(Ns * info["nd"]) - 0.5).astype(dt), (rng.random(Ns * info["k0"]) - 0.5).astype(dt)
    assert sctl_amd.plan(name, 0 if f64 else 1, Nt, Ns, digits=digits)["path"] == "tile-centred", (name, dt.__name__, digits)
    if rng.random() < 0.5:
        u = sctl_amd.eval_host(name, xt, xs, xn, f, digits=digits, devices=[0] * int(rng.integers(1, 4)))
    else:
        op = sctl_amd.DirectOp(name, dt, devices=[0] * int(rng.integers(1, 4)))
        op.set_targets(xt); op.set_sources(xs, xn)
        u = op.eval(f, digits=digits)
        op.close()
    ref = O.eval(name, xt.astype(np.float64), xs.astype(np.float64), xn.astype(np.float64), f.astype(np.float64))
    tol = (1e-12 if digits < 0 or digits >= 15 else 10.0 * 10.0 ** -digits) if f64 else (1e-4 if digits == 5 else 3e-5)
    note("forced tile-centred", rel(u, ref), tol, (c, name, dt.__name__, digits, Nt, Ns, kind, scale))
del os.environ["SCTL_AMD_CENTERED"]
print("forced tile-centred: %d cases, worst err/tol %.3f  (%.0f s)" % (n_cases, worst.get("forced tile-centred", 0), time.time() - t0), flush=True)

t0 = time.time()
for c in range(max(1, n_cases // 5)):                       # ---- fused far + near potential on random operators
    rng = np.random.default_rng(70000 + c)
    name = ["Laplace3D-FxU", "Stokes3D-FxU", "Laplace3D-DxU", "Stokes3D-DxU"][int(rng.integers(0, 4))]
    info = sctl_amd.kernel_info(name)
    k0, k1 = info["k0"], info["k1"]
    nelem, ntrg = int(rng.integers(1, 200)), int(rng.integers(1, 3000))
    nds, near = rng.integers(0, 7, nelem), rng.integers(0, 60, nelem)
    kcnt = nds * near
    kcnt[rng.random(nelem) < 0.1] = 0
    K = rng.standard_normal(int(kcnt.sum()) * k0 * k1)
    n_near = int(near.sum())
    trg_of_entry = rng.integers(0, ntrg, n_near)
    order = np.argsort(trg_of_entry, kind="stable")
    cnt = np.bincount(trg_of_entry, minlength=ntrg)
    dsp = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    Nsrc = int(nds.sum())
    if Nsrc == 0:
        continue
    xt, xs, xn = rng.random(ntrg * 3), rng.random(Nsrc * 3) + 2.0, rng.random(Nsrc * info["nd"]) - 0.5
    w, F = rng.random(Nsrc), rng.standard_normal(Nsrc * k0)
    op = sctl_amd.DirectOp(name, np.float64, devices=[0] * int(rng.integers(1, 4)))
    op.set_targets(xt); op.set_sources(xs, xn); op.set_source_weights(w)
    op.set_near(k1, nds, near, K, order, cnt, dsp, K_near_cnt=kcnt)
    u = op.eval_potential(F, F)
    op.close()
    far = O.eval(name, xt, xs, xn, (F.reshape(-1, k0) * w[:, None]).ravel())
    nearu = oracle.near_apply_restatement(k0, k1, nds, near, kcnt, K, order, cnt, dsp, F)
    note("fused potential", rel(u, far + nearu), 1e-12, (c, name, nelem, ntrg))
print("fused potential: %d cases, worst err/tol %.3f  (%.0f s)" % (max(1, n_cases // 5), worst.get("fused potential", 0), time.time() - t0), flush=True)
print("ALL OK")
