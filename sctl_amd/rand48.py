"""Bit-exact POSIX drand48 (48-bit LCG) so that inputs of the golden fixtures are regenerated, not stored.

The reference's own driver draws every input from drand48 after srand48(rank)
(/root/reference/include/sctl/fmm-wrapper.txx:41-55).  X_{n+1} = (a X_n + c) mod 2^48 with
a = 0x5DEECE66D, c = 0xB; srand48(seed) sets X = (seed << 16) | 0x330E; drand48 = X_{n+1} / 2^48.
"""
import numpy as np

_A = 0x5DEECE66D
_C = 0xB
_M = (1 << 48) - 1


_BLK = 1 << 16
_JUMP = None


def _jump_tables():
    """Affine maps of 1.._BLK steps: x_{n+i} = (A_i x_n + C_i) mod 2^48.  uint64 products wrap mod 2^64, which keeps
    the low 48 bits exact, so the whole block is one vector multiply-add."""
    global _JUMP
    if _JUMP is None:
        ak = np.empty(_BLK, dtype=np.uint64)
        ck = np.empty(_BLK, dtype=np.uint64)
        a, c = 1, 0
        for i in range(_BLK):
            a, c = (a * _A) & _M, (c * _A + _C) & _M
            ak[i], ck[i] = a, c
        _JUMP = (ak, ck)
    return _JUMP


class Rand48:
    def __init__(self, seed=0):
        self.x = ((int(seed) & 0xFFFFFFFF) << 16) | 0x330E

    def drand48(self, n):
        """n successive drand48() values as float64 (exact: 48-bit integers scaled by 2^-48)."""
        out = np.empty(n, dtype=np.float64)
        x = self.x
        if n >= 2048:
            ak, ck = _jump_tables()
            mask = np.uint64(_M)
            done = 0
            while done < n:
                m = min(_BLK, n - done)
                with np.errstate(over="ignore"):
                    vals = (ak[:m] * np.uint64(x) + ck[:m]) & mask
                out[done:done + m] = vals
                x = int(vals[-1])
                done += m
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
