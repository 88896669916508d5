"""Bit-exact POSIX drand48 (48-bit LCG) so that inputs of the golden fixtures are regenerated, not stored.

The reference's own driver draws every input from drand48 after srand48(rank)
(/root/reference/include/sctl/fmm-wrapper.txx:41-55).  X_{n+1} = (a X_n + c) mod 2^48 with
a = 0x5DEECE66D, c = 0xB; srand48(seed) sets X = (seed << 16) | 0x330E; drand48 = X_{n+1} / 2^48.
"""
import numpy as np

_A = 0x5DEECE66D
_C = 0xB
_M = (1 << 48) - 1


class Rand48:
    def __init__(self, seed=0):
        self.x = ((int(seed) & 0xFFFFFFFF) << 16) | 0x330E

    def drand48(self, n):
        """n successive drand48() values as float64 (exact: 48-bit integers scaled by 2^-48)."""
        out = np.empty(n, dtype=np.float64)
        x = self.x
        # jump-ahead by blocks: the LCG is affine, so k steps are x -> (A_k x + C_k) mod 2^48
        blk = 4096
        if n >= 4 * blk:
            # per-lane affine maps for offsets 1..blk, then stride over blocks
            ak = np.empty(blk, dtype=object)
            ck = np.empty(blk, dtype=object)
            a, c = 1, 0
            for i in range(blk):
                a, c = (a * _A) & _M, (c * _A + _C) & _M
                ak[i], ck[i] = a, c
            done = 0
            while n - done >= blk:
                vals = [(int(ak[i]) * x + int(ck[i])) & _M for i in range(blk)]
                out[done:done + blk] = np.asarray(vals, dtype=np.float64)
                x = vals[-1]
                done += blk
            for i in range(done, n):
                x = (_A * x + _C) & _M
                out[i] = x
        else:
            for i in range(n):
                x = (_A * x + _C) & _M
                out[i] = x
        self.x = x
        return out * (1.0 / (1 << 48))


def point_cloud(seed, Nt, Ns, k0, nd, dtype=np.float64, shift=-0.5):
    """Inputs in the reference driver's generation order: targets, sources, normals, densities
    (fmm-wrapper.txx:45-55), each value drand48()+shift, cast to dtype."""
    g = Rand48(seed)
    xt = (g.drand48(Nt * 3) + shift).astype(dtype)
    xs = (g.drand48(Ns * 3) + shift).astype(dtype)
    xn = (g.drand48(Ns * nd) - 0.5).astype(dtype)
    f = (g.drand48(Ns * k0) - 0.5).astype(dtype)
    return xt, xs, xn, f
