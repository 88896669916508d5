"""Per-pair error of the device Helmholtz kernel: many targets against ONE active source (the other sources carry zero density), so every
output is a single kernel value; compared with numpy long double.  Both tile passes: Ns = 1 runs the careful (masked) pass, Ns = 2048 the
speculative one.    python tools/helmholtz_pair_error.py [kr ki ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sctl_amd
L = np.longdouble
PI = L("3.14159265358979323846264338327950288")
ks = [(7.5, 0.3), (40.0, 10.0), (40.0, -10.0), (40.0, 0.0), (900.0, 5.0), (3.0, 60.0)] if len(sys.argv) < 3 else [(float(sys.argv[1]), float(sys.argv[2]))]
rng = np.random.default_rng(1)
Nt = 200000
r_t = np.sort(rng.random(Nt)) * 1.7 + 1e-3
dirs = rng.standard_normal((Nt, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
xt = (dirs * r_t[:, None]).ravel().copy()
for k in ks:
    for Ns in (1, 2048):
        xs = np.zeros(Ns * 3); xs[3:] = rng.random((Ns - 1) * 3) + 5.0          # the active source at the origin, the rest far away with zero density
        f = np.zeros(Ns * 2); f[0], f[1] = 1.0, 0.0
        u = sctl_amd.eval_host("Helmholtz3D-FxU", xt, xs, None, f, ctx=np.array(k)).reshape(Nt, 2)
        d = xt.reshape(Nt, 3).astype(L)
        r = np.sqrt((d * d).sum(-1))
        amp = np.exp(-L(k[1]) * r) / (4 * PI * r)
        tr, ti = amp * np.cos(L(k[0]) * r), amp * np.sin(L(k[0]) * r)
        err = (np.hypot((u[:, 0] - tr).astype(np.float64), (u[:, 1] - ti).astype(np.float64)) / amp.astype(np.float64))
        bins = np.linspace(0, 1.7, 6)
        per = ["%.1e" % err[(r_t >= a) & (r_t < b)].max() for a, b in zip(bins[:-1], bins[1:])]
        cond = (abs(k[0]) + abs(k[1])) * r_t                                          # what one ulp of the distance does to e^{ikr}
        print("      against the conditioning bound 1e-15 + 4e-16 |k| r: worst ratio %.2f" % (err / (1e-15 + 4e-16 * cond)).max())
        print("k = %g%+gi  Ns = %4d (%s pass): max |err|/|G| %.2e, rms %.2e; by distance fifth: %s" % (k[0], k[1], Ns, "careful" if Ns == 1 else "speculative", err.max(), np.sqrt((err ** 2).mean()), " ".join(per)), flush=True)
