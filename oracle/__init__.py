"""TEST INFRASTRUCTURE ONLY — ctypes access to the CPU oracle and to the compiled reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under sctl_amd/ or include/ does (tests/test_boundary.py checks that).

  oracle.restatement()  -> Oracle     (oracle/libsctl_oracle.so, built from oracle/sctl_oracle.cpp)
  oracle.reference()    -> Reference  (oracle/_ref/libsctl_ref_<isa>.so, the real SCTL headers; None if absent)
  oracle.build()        -> compiles both (the reference only where /root/reference exists)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KERNELS = ["Laplace3D-FxU", "Laplace3D-DxU", "Laplace3D-FxdU", "Stokes3D-FxU", "Stokes3D-DxU", "Stokes3D-FxT",
           "Stokes3D-FSxU", "Stokes3D-FxUP", "Laplace3D-FDxUdU", "Helmholtz3D-FxU"]


def build(ref=True):
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref and os.path.isdir("/root/reference/include/sctl"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref", "-j2"], check=True)


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def _ctx(ctx):
    if ctx is None:
        return None, None
    buf = np.ascontiguousarray(ctx, dtype=np.float64)
    return buf, buf.ctypes.data_as(C.c_void_p)


class _Base:
    def info(self, name):
        k0, k1, nd, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        sc = C.c_double()
        rc = self._info(name.encode(), C.byref(k0), C.byref(k1), C.byref(nd), C.byref(fl), C.byref(sc))
        if rc != 0:
            raise KeyError(name)
        return dict(k0=k0.value, k1=k1.value, nd=nd.value, flops=fl.value, scale=sc.value)


class Oracle(_Base):
    """CPU restatement (exact sqrt/divide).  eval() ACCUMULATES into v_trg like generic-kernel.txx:184."""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self._info = self.lib.sctl_oracle_kernel_info
        for f in (self.lib.sctl_oracle_eval_f64, self.lib.sctl_oracle_eval_f32):
            f.argtypes = [C.c_char_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_int]
            f.restype = C.c_int
        for f in (self.lib.sctl_oracle_matrix_f64, self.lib.sctl_oracle_matrix_f32):
            f.argtypes = [C.c_char_p, C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [C.c_int]
            f.restype = C.c_int
        self.kind = "port"

    def num_threads(self):
        return self.lib.sctl_oracle_num_threads()

    def eval(self, name, xt, xs, xn, f, v_trg=None, ctx=None, nthreads=0, digits=-1):
        dt = xt.dtype
        inf = self.info(name)
        Nt, Ns = xt.size // 3, xs.size // 3
        assert xs.dtype == dt and f.dtype == dt and f.size == Ns * inf["k0"] and (inf["nd"] == 0 or xn.size == Ns * inf["nd"])
        if v_trg is None:
            v_trg = np.zeros(Nt * inf["k1"], dtype=dt)
        assert v_trg.size == Nt * inf["k1"] and v_trg.dtype == dt
        fn = self.lib.sctl_oracle_eval_f64 if dt == np.float64 else self.lib.sctl_oracle_eval_f32
        keep, cp = _ctx(ctx)
        rc = fn(name.encode(), Nt, Ns, _ptr(xt), _ptr(xs), _ptr(xn), _ptr(f), _ptr(v_trg), cp, nthreads)
        assert rc == 0, rc
        return v_trg

    def kernel_matrix(self, name, xt, xs, xn, ctx=None, nthreads=0):
        dt = xt.dtype
        inf = self.info(name)
        Nt, Ns = xt.size // 3, xs.size // 3
        M = np.zeros((Ns * inf["k0"], Nt * inf["k1"]), dtype=dt)
        fn = self.lib.sctl_oracle_matrix_f64 if dt == np.float64 else self.lib.sctl_oracle_matrix_f32
        keep, cp = _ctx(ctx)
        rc = fn(name.encode(), Nt, Ns, _ptr(xt), _ptr(xs), _ptr(xn), _ptr(M), cp, nthreads)
        assert rc == 0, rc
        return M


_REAL = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.longdouble): 2}


class Reference(_Base):
    """The real reference (SCTL GenericKernel / ParticleFMM) behind oracle/ref_shim.cpp."""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self._info = self.lib.sctl_ref_kernel_info
        self.lib.sctl_ref_eval.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int]
        self.lib.sctl_ref_kernel_matrix.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 5
        self.lib.sctl_ref_particle_fmm_eval_direct.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [C.c_int]
        self.lib.sctl_ref_isa.restype = C.c_char_p
        self.kind = "reference"
        self.isa = self.lib.sctl_ref_isa().decode()

    def num_threads(self):
        return self.lib.sctl_ref_num_threads()

    def eval(self, name, xt, xs, xn, f, v_trg=None, ctx=None, digits=-1, omp=True, nthreads=0):
        dt = xt.dtype
        inf = self.info(name)
        Nt, Ns = xt.size // 3, xs.size // 3
        if v_trg is None:
            v_trg = np.zeros(Nt * inf["k1"], dtype=dt)
        assert v_trg.size == Nt * inf["k1"] and v_trg.dtype == dt
        keep, cp = _ctx(ctx)
        rc = self.lib.sctl_ref_eval(name.encode(), _REAL[np.dtype(dt)], Nt, Ns, _ptr(xt), _ptr(xs), _ptr(xn), _ptr(f), _ptr(v_trg),
                                    digits, cp, 1 if omp else 0)
        assert rc == 0, rc
        return v_trg

    def kernel_matrix(self, name, xt, xs, xn, ctx=None):
        dt = xt.dtype
        inf = self.info(name)
        Nt, Ns = xt.size // 3, xs.size // 3
        M = np.zeros((Ns * inf["k0"], Nt * inf["k1"]), dtype=dt)
        keep, cp = _ctx(ctx)
        rc = self.lib.sctl_ref_kernel_matrix(name.encode(), _REAL[np.dtype(dt)], Nt, Ns, _ptr(xt), _ptr(xs), _ptr(xn), _ptr(M), cp)
        assert rc == 0, rc
        return M

    def boundary_far_field(self, name, xt, xn_trg, xs, xn, wts, f, trg_normal_dot_prod=False, tol=1e-10, nodes_per_elem=1, upsample=1):
        """BoundaryIntegralOp::ComputePotential on a point element list with no near zone == ComputeFarField
        (boundary_integral.txx:1016-1077).  xt=None: targets are the surface nodes."""
        inf = self.info(name)
        Ns = xs.size // 3
        Nt = 0 if xt is None else xt.size // 3
        NT = Nt if Nt else Ns
        k1 = inf["k1"] // 3 if trg_normal_dot_prod else inf["k1"]
        U = np.zeros(NT * k1, dtype=np.float64)
        n = C.c_int64()
        fn = self.lib.sctl_ref_boundary_far_field
        fn.argtypes = [C.c_char_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        rc = fn(name.encode(), Nt, Ns, _ptr(xt), _ptr(xn_trg), _ptr(xs), _ptr(xn), _ptr(wts), _ptr(f), 1 if trg_normal_dot_prod else 0,
                tol, nodes_per_elem, upsample, _ptr(U), C.byref(n))
        assert rc == 0 and n.value == U.size, (rc, n.value, U.size)
        return U

    def particle_fmm_eval_direct(self, name, xt, xs, xn, f, digits=10):
        inf = self.info(name)
        Nt, Ns = xt.size // 3, xs.size // 3
        U = np.zeros(Nt * inf["k1"], dtype=np.float64)
        rc = self.lib.sctl_ref_particle_fmm_eval_direct(name.encode(), 0, Nt, Ns, _ptr(xt), _ptr(xs), _ptr(xn), _ptr(f), _ptr(U), digits)
        assert rc == 0, rc
        return U


def far_field_restatement(O, name, xt, xn_trg, xs, xn, wts, f, trg_normal_dot_prod=False):
    """CPU restatement of BoundaryIntegralOp::ComputeFarField (boundary_integral.txx:1016-1077): density x weights,
    direct sum, optional dot product of the K1/3 x 3 output with the target normals.  xt=None: targets = surface nodes."""
    inf = O.info(name)
    Ns = xs.size // 3
    T, Tn = (xs, xn) if xt is None else (xt, xn_trg)
    NT = T.size // 3
    fw = (f.reshape(Ns, inf["k0"]) * wts[:, None]).ravel()
    u = O.eval(name, T, xs, xn if inf["nd"] else None, fw)
    if trg_normal_dot_prod:
        u = (u.reshape(NT, inf["k1"] // 3, 3) * Tn.reshape(NT, 1, 3)).sum(-1).ravel()
    return u


def restatement():
    path = os.path.join(_HERE, "libsctl_oracle.so")
    if not os.path.exists(path):
        build(ref=False)
    return Oracle(path)


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def reference():
    """The compiled reference for this host's ISA, or None when oracle/_ref holds no usable build."""
    order = (["avx512"] if _cpu_has("avx512f") and _cpu_has("avx512dq") and _cpu_has("avx512vl") and _cpu_has("avx512bw") else []) + ["avx2"]
    for isa in order:
        path = os.path.join(_HERE, "_ref", "libsctl_ref_%s.so" % isa)
        if os.path.exists(path) and (isa != "avx2" or _cpu_has("avx2")):
            return Reference(path)
    return None


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))
