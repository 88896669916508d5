"""TEST INFRASTRUCTURE ONLY — ctypes access to the CPU oracle and to the compiled reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under sctl_amd/ or include/ does (tests/test_boundary.py checks that).

  oracle.restatement()  -> Oracle     (oracle/libsctl_oracle.so, built from oracle/sctl_oracle.cpp)
  oracle.reference()    -> Reference  (oracle/_ref/libsctl_ref_<isa>.so, the real SCTL headers; None if absent)
  oracle.dropin()       -> Reference  (the same shim built with include/sctl_amd/sctl_dropin.hpp's kernel class: the reference's call
                                       sites running on libsctl_amd.so; the thing under test in tests/test_gpu_dropin.py)
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
        subprocess.run(["make", "-s", "-C", _HERE, "ref", "-j5"], check=True)


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

    def set_devices(self, devices=None, min_pairs_per_device=-1):
        """Drop-in build only: the GPU list of sctl_dropin.hpp (sctl_amd::Devices()) and its work-per-GPU threshold."""
        return set_dropin_devices(self.lib, devices, min_pairs_per_device)

    def get_devices(self):
        buf = (C.c_int * 64)()
        mp = C.c_longlong()
        n = self.lib.sctl_ref_dropin_get_devices(buf, 64, C.byref(mp))
        return list(buf[:max(n, 0)]), mp.value

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


def set_dropin_devices(lib, devices=None, min_pairs_per_device=-1):
    lib.sctl_ref_dropin_set_devices.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
    d = np.ascontiguousarray(devices if devices is not None else [], dtype=np.int32)
    n = lib.sctl_ref_dropin_set_devices(_ptr(d), d.size, min_pairs_per_device)
    assert n > 0, "not a drop-in build"
    return n


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


def dropin():
    """The reference's own call sites (GenericKernel::Eval entries, ParticleFMM, BoundaryIntegralOp) compiled with the kernel class of
    include/sctl_amd/sctl_dropin.hpp, so that their arithmetic runs in libsctl_amd.so: oracle/ref_shim.cpp with -DSCTL_REF_DROPIN.
    Needs a GPU to evaluate anything.  None if oracle/_ref holds no such build."""
    path = os.path.join(_HERE, "_ref", "libsctl_ref_dropin.so")
    return Reference(path) if os.path.exists(path) else None


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


# ---- BoundaryIntegralOp near field (boundary_integral.txx:784-1012 setup, :1079-1142 apply) ---------------------------------
NEAR_INT_ARRAYS = ("elem_nds_cnt", "near_elem_cnt", "K_near_cnt", "near_scatter_index", "near_trg_cnt", "near_trg_dsp")


def reference_near(name, xt, xn_trg, xs, xn, wts, f, trg_normal_dot_prod=False, tol=1e-10, nodes_per_elem=4, upsample=1, rad=0.1, free_nodes=0, dropin=False,
                   dropin_devices=None):
    """The REAL reference's BoundaryIntegralOp on the synthetic PatchElemList of oracle/ref_near_shim.cpp (build container only).
    Returns u_total (ComputePotential), u_near (ComputeNearInterac alone) and the near-operator arrays SetupNear built.
    free_nodes > 0: TWO element lists — the last `free_nodes` nodes form a second, matrix-free list (sctl_ref_boundary_near2)."""
    path = os.path.join(_HERE, "_ref", "libsctl_ref_near_dropin.so" if dropin else "libsctl_ref_near.so")   # dropin: kernels = sctl_amd::HipKernel
    lib = C.CDLL(path)
    if dropin and dropin_devices is not None:
        set_dropin_devices(lib, dropin_devices, 0)
    Ns = xs.size // 3
    Nt = 0 if xt is None else xt.size // 3
    NT = Nt if Nt else Ns
    ns_a = Ns - free_nodes
    nelem = (ns_a + nodes_per_elem - 1) // nodes_per_elem + (free_nodes + nodes_per_elem - 1) // nodes_per_elem
    near_cap, k_cap = NT * nelem, NT * nelem * nodes_per_elem * 9
    out = dict(u_total=np.zeros(NT * 9), u_near=np.zeros(NT * 9), K_near=np.zeros(k_cap))
    for k, n in (("elem_nds_cnt", nelem), ("near_elem_cnt", nelem), ("K_near_cnt", nelem), ("near_scatter_index", near_cap), ("near_trg_cnt", NT), ("near_trg_dsp", NT)):
        out[k] = np.zeros(n, dtype=np.int64)
    sizes = np.zeros(8, dtype=np.int64)
    fn = lib.sctl_ref_boundary_near2 if free_nodes else lib.sctl_ref_boundary_near
    fn.argtypes = ([C.c_char_p, C.c_int64, C.c_int64] + ([C.c_int64] if free_nodes else []) + [C.c_void_p] * 6 + [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double] +
                   [C.c_void_p] * 2 + [C.c_int64] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64])
    sizes_in = (Nt, ns_a, free_nodes) if free_nodes else (Nt, Ns)
    rc = fn(name.encode(), *sizes_in, _ptr(xt), _ptr(xn_trg), _ptr(xs), _ptr(xn), _ptr(wts), _ptr(f), 1 if trg_normal_dot_prod else 0, tol, nodes_per_elem,
            upsample, rad, _ptr(out["u_total"]), _ptr(out["u_near"]), out["u_total"].size, _ptr(sizes), _ptr(out["elem_nds_cnt"]),
            _ptr(out["near_elem_cnt"]), _ptr(out["K_near_cnt"]), nelem, _ptr(out["near_scatter_index"]), near_cap, _ptr(out["near_trg_cnt"]),
            _ptr(out["near_trg_dsp"]), NT, _ptr(out["K_near"]), k_cap)
    assert rc == 0, rc
    ntrg, ne, nnear, klen, ulen = (int(v) for v in sizes[:5])
    assert ntrg == NT and ne == nelem
    out["u_total"], out["u_near"], out["K_near"] = out["u_total"][:ulen].copy(), out["u_near"][:ulen].copy(), out["K_near"][:klen].copy()
    out["near_scatter_index"] = out["near_scatter_index"][:nnear].copy()
    return out


def near_apply_restatement(k0, k1, elem_nds_cnt, near_elem_cnt, K_near_cnt, K_near, near_scatter_index, near_trg_cnt, near_trg_dsp, F, U=None):
    """CPU restatement of BoundaryIntegralOp::ComputeNearInterac (boundary_integral.txx:1079-1142) for precomputed operator
    matrices: per element U_ = F_ . K_near_ (:1092-1102), permutation by near_scatter_index (:1129, ScatterForward:
    out[i] = in[index[i]]), per target the sum of its near_trg_cnt entries in order (:1131-1140), ACCUMULATED into U.
    k1 is the number of potential components per target (already divided by 3 under trg_normal_dot_prod)."""
    nelem, ntrg = elem_nds_cnt.size, near_trg_cnt.size
    nds_dsp = np.concatenate([[0], np.cumsum(elem_nds_cnt)])
    near_dsp = np.concatenate([[0], np.cumsum(near_elem_cnt)])
    k_dsp = np.concatenate([[0], np.cumsum(K_near_cnt)])
    u_near = np.zeros(int(near_dsp[-1]) * k1, dtype=K_near.dtype)
    for e in range(nelem):
        sd, td = int(elem_nds_cnt[e]) * k0, int(near_elem_cnt[e]) * k1
        if sd == 0 or td == 0 or K_near_cnt[e] == 0:
            continue
        Kb = K_near[int(k_dsp[e]) * k0 * k1:int(k_dsp[e]) * k0 * k1 + sd * td].reshape(sd, td)
        Fe = F[int(nds_dsp[e]) * k0:int(nds_dsp[e]) * k0 + sd]
        acc = np.zeros(td, dtype=K_near.dtype)
        for s in range(sd):                     # row by row, the order of a row-major GEMV
            acc += Fe[s] * Kb[s]
        u_near[int(near_dsp[e]) * k1:int(near_dsp[e]) * k1 + td] = acc
    u_sc = u_near.reshape(-1, k1)[near_scatter_index]
    if U is None:
        U = np.zeros(ntrg * k1, dtype=K_near.dtype)
    Uv = U.reshape(ntrg, k1)
    for i in range(ntrg):
        for p in range(int(near_trg_dsp[i]), int(near_trg_dsp[i]) + int(near_trg_cnt[i])):
            Uv[i] += u_sc[p]
    return U
