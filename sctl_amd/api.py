"""ctypes binding of include/sctl_amd.h and a GenericKernel mirror of the reference's kernel objects.

Names, argument order and semantics follow the reference (file:line relative to /root/reference):
  GenericKernel.Eval(v_trg, r_trg, r_src, n_src, v_src)   include/sctl/generic-kernel.hpp:123, generic-kernel.txx:76-189
  GenericKernel.KernelMatrix(M, Xt, Xs, Xn)               include/sctl/generic-kernel.hpp:135, generic-kernel.txx:191-307
  SrcDim / TrgDim / NormalDim / CoordDim / Name / FLOPS   include/sctl/generic-kernel.hpp:59-84, kernel_functions.hpp:16-22
numpy arrays go through the host-buffer entry points, torch CUDA tensors through the device-resident ones
(on torch's current stream).  A missing or failing library raises; nothing is computed on the CPU here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F64, F32 = 0, 1
KERNEL_NAMES = ["Laplace3D-FxU", "Laplace3D-DxU", "Laplace3D-FxdU", "Stokes3D-FxU", "Stokes3D-DxU", "Stokes3D-FxT", "Stokes3D-FSxU",
                "Stokes3D-FxUP", "Laplace3D-FDxUdU", "Helmholtz3D-FxU"]

# every symbol include/sctl_amd.h declares (tests/test_boundary.py checks the header against this list and the .so)
SYMBOLS = ["sctl_amd_version", "sctl_amd_last_error", "sctl_amd_device_count", "sctl_amd_init", "sctl_amd_finalize", "sctl_amd_kernel_id", "sctl_amd_kernel_name",
           "sctl_amd_kernel_info", "sctl_amd_flops_per_pair", "sctl_amd_eval_device", "sctl_amd_eval_device_slab", "sctl_amd_eval_host", "sctl_amd_eval_host_multi",
           "sctl_amd_kernel_matrix_device", "sctl_amd_kernel_matrix_host", "sctl_amd_kernel_matrix_batch_host", "sctl_amd_counters", "sctl_amd_reset_counters", "sctl_amd_trim",
           "sctl_amd_eval_plan", "sctl_amd_eval_path", "sctl_amd_eval_pipe", "sctl_amd_op_create", "sctl_amd_op_set_targets",
           "sctl_amd_op_set_sources", "sctl_amd_op_set_source_weights", "sctl_amd_op_set_target_normals", "sctl_amd_op_eval", "sctl_amd_op_destroy", "sctl_amd_near_create", "sctl_amd_near_apply_host",
           "sctl_amd_near_apply_device", "sctl_amd_near_info", "sctl_amd_near_destroy", "sctl_amd_num_kernels", "sctl_amd_register_kernel", "sctl_amd_load_plugin",
           "sctl_amd_set_debug", "sctl_amd_comm_create", "sctl_amd_comm_info", "sctl_amd_comm_allgatherv_host", "sctl_amd_comm_barrier", "sctl_amd_comm_selftest", "sctl_amd_comm_destroy",
           "sctl_amd_op_set_sources_dist", "sctl_amd_op_eval_dist", "sctl_amd_op_set_near", "sctl_amd_op_eval_potential", "sctl_amd_lists_create", "sctl_amd_lists_eval_device", "sctl_amd_lists_eval_host", "sctl_amd_lists_info", "sctl_amd_lists_destroy",
           "sctl_amd_eval_lists_device", "sctl_amd_eval_lists_host"]


class SctlAmdError(RuntimeError):
    pass


def library_path():
    # SCTL_AMD_LIB: explicit path of another build of the same library (kernel experiments under tools/)
    return os.environ.get("SCTL_AMD_LIB") or os.path.join(_HERE, "libsctl_amd.so")


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (soname libamdhip64.so.7, found through its
    RPATH); libsctl_amd.so asks for libamdhip64.so.7 and, loaded first, gets /opt/rocm's copy — a later `import torch` then
    loads a SECOND runtime under the other file name and finds no GPU ("No HIP GPUs are available").  Loading torch's copy
    first makes both resolve to it, whichever side is used first.  Skipped when torch is absent or already imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("SCTL_AMD_OWN_HIP_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise SctlAmdError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                           "`make -C sctl_amd/csrc` (hipcc, gfx950); sctl_amd has no CPU fallback" % path)
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    L.sctl_amd_last_error.restype = C.c_char_p
    L.sctl_amd_kernel_name.restype = C.c_char_p
    L.sctl_amd_kernel_id.argtypes = [C.c_char_p]
    L.sctl_amd_kernel_info.argtypes = [ci] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.sctl_amd_eval_device.argtypes = [ci, ci, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, vp]
    L.sctl_amd_eval_device_slab.argtypes = [ci, ci, i64, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, vp]
    L.sctl_amd_eval_host.argtypes = [ci, ci, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, ci]
    L.sctl_amd_eval_host_multi.argtypes = [ci, ci, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, C.POINTER(C.c_int), ci]
    L.sctl_amd_kernel_matrix_device.argtypes = [ci, ci, i64, i64, vp, vp, vp, vp, ci, vp, ci, vp]
    L.sctl_amd_kernel_matrix_host.argtypes = [ci, ci, i64, i64, vp, vp, vp, vp, ci, vp, ci, ci]
    L.sctl_amd_kernel_matrix_batch_host.argtypes = [ci, ci, i64, vp, vp, vp, vp, vp, vp, ci, vp, ci, ci]
    L.sctl_amd_counters.argtypes = [C.POINTER(i64), C.POINTER(i64)]
    L.sctl_amd_counters.restype = None
    L.sctl_amd_reset_counters.restype = None
    L.sctl_amd_trim.restype = None
    pi64 = C.POINTER(i64)
    L.sctl_amd_near_create.argtypes = [ci, ci, i64, ci, ci, vp, vp, vp, vp, i64, vp, vp, vp, C.POINTER(vp)]
    L.sctl_amd_near_apply_host.argtypes = [vp, vp, vp]
    L.sctl_amd_near_apply_device.argtypes = [vp, vp, vp, vp]
    L.sctl_amd_near_info.argtypes = [vp, pi64, pi64, pi64, pi64, pi64]
    L.sctl_amd_near_destroy.argtypes = [vp]
    L.sctl_amd_near_destroy.restype = None
    L.sctl_amd_eval_plan.argtypes = [ci, ci, i64, i64, i64, ci, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64)]
    L.sctl_amd_eval_path.argtypes = [ci, ci, i64, i64, i64]
    L.sctl_amd_eval_pipe.argtypes = [ci, ci, i64, i64, i64, ci]
    L.sctl_amd_op_create.argtypes = [ci, ci, C.POINTER(C.c_int), ci, C.POINTER(vp)]
    L.sctl_amd_op_set_targets.argtypes = [vp, i64, vp]
    L.sctl_amd_op_set_sources.argtypes = [vp, i64, vp, vp]
    L.sctl_amd_op_set_source_weights.argtypes = [vp, vp]
    L.sctl_amd_op_set_target_normals.argtypes = [vp, vp]
    L.sctl_amd_op_eval.argtypes = [vp, vp, vp, ci, ci, vp, ci]
    L.sctl_amd_op_destroy.argtypes = [vp]
    L.sctl_amd_op_destroy.restype = None
    L.sctl_amd_op_set_near.argtypes = [vp, ci, ci, i64, vp, vp, vp, vp, vp, vp, vp]
    L.sctl_amd_op_eval_potential.argtypes = [vp, vp, vp, vp, ci, ci, vp, ci]
    L.sctl_amd_load_plugin.argtypes = [C.c_char_p]
    L.sctl_amd_finalize.restype = None
    L.sctl_amd_lists_create.argtypes = [ci, ci, ci, i64, vp, vp, vp, vp, i64, i64, C.POINTER(vp)]
    L.sctl_amd_lists_eval_device.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp, ci, vp]
    L.sctl_amd_lists_eval_host.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp, ci]
    L.sctl_amd_lists_info.argtypes = [vp, pi64, pi64, pi64]
    L.sctl_amd_lists_destroy.argtypes = [vp]
    L.sctl_amd_lists_destroy.restype = None
    L.sctl_amd_eval_lists_host.argtypes = [ci, ci, i64, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, ci]
    L.sctl_amd_eval_lists_device.argtypes = [ci, ci, i64, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, ci, vp, ci, vp]
    _LIB = L
    return L


def last_error():
    return lib().sctl_amd_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise SctlAmdError("%s failed with status %d: %s" % (what, rc, last_error()))


def init():
    """Touch every visible GPU now (optional): returns their number."""
    n = lib().sctl_amd_init()
    if n < 0:
        raise SctlAmdError("init failed with status %d: %s" % (n, last_error()))
    return n


def finalize():
    """Give back everything the library keeps between calls (optional; it stays usable)."""
    lib().sctl_amd_finalize()


def device_count():
    return lib().sctl_amd_device_count()


def kernel_id(name):
    """Device kernel id for a functor Name(), or raises KeyError (the 'is it supported' query)."""
    k = name if isinstance(name, int) else lib().sctl_amd_kernel_id(name.encode())
    if k < 0 or k >= lib().sctl_amd_num_kernels():
        raise KeyError("kernel %r is not implemented on the device" % (name,))
    return k


def load_plugin(path):
    """dlopen a kernel plugin (a user functor compiled against include/sctl_amd/device/kernel_plugin.hpp): returns the names it registered."""
    n0 = lib().sctl_amd_num_kernels()
    rc = lib().sctl_amd_load_plugin(os.fspath(path).encode())
    if rc < 0:
        raise SctlAmdError("load_plugin(%s) failed with status %d: %s" % (path, rc, last_error()))
    return [lib().sctl_amd_kernel_name(i).decode() for i in range(n0, n0 + rc)]


_INFO_CACHE = {}


def kernel_info(name):
    """Shape table of a kernel (by Name() or id).  Cached: a kernel's entry never changes once registered, and the evaluation wrappers
    ask for it on every call (three ctypes round trips otherwise — visible in the step time of 2^14-point problems)."""
    hit = _INFO_CACHE.get(name)
    if hit is not None:
        return hit
    info = _kernel_info_uncached(name)
    _INFO_CACHE[name] = _INFO_CACHE[info["id"]] = _INFO_CACHE[info["name"]] = info
    return info


def _kernel_info_uncached(name):
    k = kernel_id(name)
    v = [C.c_int() for _ in range(4)]
    sc, cb = C.c_double(), C.c_int()
    _check(lib().sctl_amd_kernel_info(k, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), C.byref(sc), C.byref(cb)), "kernel_info")
    return dict(id=k, name=lib().sctl_amd_kernel_name(k).decode(), k0=v[0].value, k1=v[1].value, nd=v[2].value, flops=v[3].value,
                scale=sc.value, ctx_bytes=cb.value)


def flops_per_pair(name):
    return lib().sctl_amd_flops_per_pair(kernel_id(name))


def plan(name, real, Nt, Ns, digits=-1, nt_whole=0):
    t, s = C.c_int(), C.c_int()
    wg, ws = C.c_int64(), C.c_int64()
    _check(lib().sctl_amd_eval_plan(kernel_id(name), real, Nt, Ns, nt_whole, digits, C.byref(t), C.byref(s), C.byref(wg), C.byref(ws)), "eval_plan")
    path = lib().sctl_amd_eval_path(kernel_id(name), real, Nt, Ns, nt_whole)
    pipe = lib().sctl_amd_eval_pipe(kernel_id(name), real, Nt, Ns, nt_whole, digits)
    return dict(trg_per_lane=t.value, src_splits=s.value, workgroups=wg.value, workspace_bytes=ws.value,
                path="tile-centred" if (path == 1 and pipe >= 1) else "exact",     # (eval_path answers for the default accuracy, eval_pipe for `digits`)
                pipe="bf16 matrix cores (r^2) + vector pipe (rsqrt, accumulate)" if pipe == 2 else "vector pipe")


def counters():
    p, f = C.c_int64(), C.c_int64()
    lib().sctl_amd_counters(C.byref(p), C.byref(f))
    return dict(pair_interactions=p.value, sctl_flops=f.value)


def reset_counters():
    lib().sctl_amd_reset_counters()


def trim():
    """Release the device scratch memory cached per (device, stream)."""
    lib().sctl_amd_trim()


def _ctx_blob(info, ctx):
    if info["ctx_bytes"] == 0:
        return None, None, 0
    if ctx is None:
        raise SctlAmdError("%s needs a context of %d bytes (e.g. the complex wavenumber)" % (info["name"], info["ctx_bytes"]))
    buf = np.ascontiguousarray(ctx, dtype=np.float64)
    return buf, buf.ctypes.data_as(C.c_void_p), buf.nbytes


def _real_of(dtype):
    dt = np.dtype(dtype)
    if dt == np.float64:
        return F64
    if dt == np.float32:
        return F32
    raise SctlAmdError("only float64 and float32 are supported, got %s" % dt)


def _np_ptr(a, dt, n, what):
    if n == 0:
        return None
    if a is None or a.dtype != dt or a.size != n or not a.flags["C_CONTIGUOUS"]:
        raise SctlAmdError("%s must be a contiguous %s array of %d values" % (what, np.dtype(dt).name, n))
    return a.ctypes.data_as(C.c_void_p)


def eval_host(name, r_trg, r_src, n_src, v_src, v_trg=None, digits=-1, ctx=None, device=0, devices=None):
    """GenericKernel::Eval on host (numpy) arrays.  v_trg of the right size is ACCUMULATED into
    (generic-kernel.txx:184); any other size (or None) gives a fresh zeroed result (generic-kernel.txx:98-101)."""
    info = kernel_info(name)
    dt = r_trg.dtype
    real = _real_of(dt)
    Nt, Ns = r_trg.size // 3, r_src.size // 3
    if r_trg.size != Nt * 3 or r_src.size != Ns * 3:
        raise SctlAmdError("coordinate arrays must hold 3 values per point")
    if v_trg is None or v_trg.size != Nt * info["k1"]:
        v_trg = np.zeros(Nt * info["k1"], dtype=dt)
    keep, cp, cb = _ctx_blob(info, ctx)
    args = [info["id"], real, Nt, Ns, _np_ptr(r_trg, dt, Nt * 3, "r_trg"), _np_ptr(r_src, dt, Ns * 3, "r_src"),
            _np_ptr(n_src, dt, Ns * info["nd"], "n_src"), _np_ptr(v_src, dt, Ns * info["k0"], "v_src"),
            _np_ptr(v_trg, dt, Nt * info["k1"], "v_trg"), digits, cp, cb]
    if devices is None:
        _check(lib().sctl_amd_eval_host(*args, device), "eval_host")
    else:
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().sctl_amd_eval_host_multi(*args, devs, len(devices)), "eval_host_multi")
    return v_trg


def _t_ptr(t, torch_dtype, n, what):
    if n == 0:
        return None
    if t is None or t.dtype != torch_dtype or t.numel() != n or not t.is_contiguous() or not t.is_cuda:
        raise SctlAmdError("%s must be a contiguous CUDA tensor of %d %s values" % (what, n, torch_dtype))
    return C.c_void_p(t.data_ptr())


def eval_device(name, r_trg, r_src, n_src, v_src, v_trg=None, digits=-1, ctx=None, stream=None, nt_whole=None):
    """GenericKernel::Eval on torch CUDA tensors, enqueued on `stream` (default: torch's current stream).
    nt_whole: the targets are a spatially compact slab of a set of that many (sctl_amd_eval_device_slab)."""
    import torch
    info = kernel_info(name)
    tdt = r_trg.dtype
    real = F64 if tdt == torch.float64 else _real_of(np.float32 if tdt == torch.float32 else np.int8)
    Nt, Ns = r_trg.numel() // 3, r_src.numel() // 3
    if v_trg is None or v_trg.numel() != Nt * info["k1"]:
        v_trg = torch.zeros(Nt * info["k1"], dtype=tdt, device=r_trg.device)
    keep, cp, cb = _ctx_blob(info, ctx)
    with torch.cuda.device(r_trg.device):
        st = stream if stream is not None else torch.cuda.current_stream()
        _check(lib().sctl_amd_eval_device_slab(info["id"], real, Nt, Ns, Nt if nt_whole is None else nt_whole, _t_ptr(r_trg, tdt, Nt * 3, "r_trg"),
                                               _t_ptr(r_src, tdt, Ns * 3, "r_src"), _t_ptr(n_src, tdt, Ns * info["nd"], "n_src"),
                                               _t_ptr(v_src, tdt, Ns * info["k0"], "v_src"), _t_ptr(v_trg, tdt, Nt * info["k1"], "v_trg"), digits, cp, cb,
                                               C.c_void_p(st.cuda_stream)), "eval_device")
    return v_trg


def kernel_matrix_host(name, r_trg, r_src, n_src, digits=-1, ctx=None, device=0):
    """GenericKernel::KernelMatrix on numpy arrays: returns M of shape (Ns*SrcDim, Nt*TrgDim), scale included."""
    info = kernel_info(name)
    dt = r_trg.dtype
    real = _real_of(dt)
    Nt, Ns = r_trg.size // 3, r_src.size // 3
    M = np.zeros((Ns * info["k0"], Nt * info["k1"]), dtype=dt)
    keep, cp, cb = _ctx_blob(info, ctx)
    _check(lib().sctl_amd_kernel_matrix_host(info["id"], real, Nt, Ns, _np_ptr(r_trg, dt, Nt * 3, "r_trg"), _np_ptr(r_src, dt, Ns * 3, "r_src"),
                                             _np_ptr(n_src, dt, Ns * info["nd"], "n_src"), _np_ptr(M, dt, M.size, "M"), digits, cp, cb, device),
           "kernel_matrix_host")
    return M


def kernel_matrix_batch_host(name, Nt, Ns, r_trg, r_src, n_src, digits=-1, ctx=None, device=0):
    """Many KernelMatrix blocks in one launch (sctl_amd_kernel_matrix_batch_host): returns the list of (Ns[b]*K0, Nt[b]*K1) blocks."""
    info = kernel_info(name)
    dt = r_trg.dtype
    real = _real_of(dt)
    Nt, Ns = np.ascontiguousarray(Nt, dtype=np.int64), np.ascontiguousarray(Ns, dtype=np.int64)
    sizes = Ns * info["k0"] * Nt * info["k1"]
    M = np.zeros(int(sizes.sum()), dtype=dt)
    keep, cp, cb = _ctx_blob(info, ctx)
    _check(lib().sctl_amd_kernel_matrix_batch_host(info["id"], real, Nt.size, Nt.ctypes.data_as(C.c_void_p), Ns.ctypes.data_as(C.c_void_p),
                                                   _np_ptr(r_trg, dt, int(Nt.sum()) * 3, "r_trg"), _np_ptr(r_src, dt, int(Ns.sum()) * 3, "r_src"),
                                                   _np_ptr(n_src, dt, int(Ns.sum()) * info["nd"], "n_src"), _np_ptr(M, dt, M.size, "M"), digits, cp, cb, device),
           "kernel_matrix_batch_host")
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [M[off[b]:off[b + 1]].reshape(int(Ns[b]) * info["k0"], int(Nt[b]) * info["k1"]) for b in range(Nt.size)]


def kernel_matrix_device(name, r_trg, r_src, n_src, M=None, digits=-1, ctx=None, stream=None):
    import torch
    info = kernel_info(name)
    tdt = r_trg.dtype
    real = F64 if tdt == torch.float64 else F32
    Nt, Ns = r_trg.numel() // 3, r_src.numel() // 3
    if M is None or M.numel() != Ns * info["k0"] * Nt * info["k1"]:
        M = torch.empty((Ns * info["k0"], Nt * info["k1"]), dtype=tdt, device=r_trg.device)
    keep, cp, cb = _ctx_blob(info, ctx)
    with torch.cuda.device(r_trg.device):
        st = stream if stream is not None else torch.cuda.current_stream()
        _check(lib().sctl_amd_kernel_matrix_device(info["id"], real, Nt, Ns, _t_ptr(r_trg, tdt, Nt * 3, "r_trg"),
                                                   _t_ptr(r_src, tdt, Ns * 3, "r_src"), _t_ptr(n_src, tdt, Ns * info["nd"], "n_src"),
                                                   _t_ptr(M, tdt, M.numel(), "M"), digits, cp, cb, C.c_void_p(st.cuda_stream)),
               "kernel_matrix_device")
    return M


class GenericKernel:
    """Python mirror of a reference kernel object (GenericKernel<uKernel>, generic-kernel.hpp:31-152)."""

    def __init__(self, name, ctx=None):
        self._info = kernel_info(name)
        self._ctx = ctx

    def Name(self):
        return self._info["name"]

    def FLOPS(self):
        return self._info["flops"]

    def uKerScaleFactor(self):
        return self._info["scale"]

    def CoordDim(self):
        return 3

    def NormalDim(self):
        return self._info["nd"]

    def SrcDim(self):
        return self._info["k0"]

    def TrgDim(self):
        return self._info["k1"]

    def SetCtxPtr(self, ctx):
        self._ctx = ctx

    def GetCtxPtr(self):
        return self._ctx

    def Eval(self, v_trg, r_trg, r_src, n_src, v_src, digits=-1, **kw):
        if isinstance(r_trg, np.ndarray):
            return eval_host(self._info["id"], r_trg, r_src, n_src, v_src, v_trg, digits, self._ctx, **kw)
        return eval_device(self._info["id"], r_trg, r_src, n_src, v_src, v_trg, digits, self._ctx, **kw)

    def KernelMatrix(self, M, Xt, Xs, Xn, digits=-1, **kw):
        if isinstance(Xt, np.ndarray):
            return kernel_matrix_host(self._info["id"], Xt, Xs, Xn, digits, self._ctx, **kw)
        return kernel_matrix_device(self._info["id"], Xt, Xs, Xn, M, digits, self._ctx, **kw)


class DirectOp:
    """Device-resident operator (sctl_amd_op_*): coordinates are uploaded once, each eval() moves only density and
    potential.  The Python face of what ParticleFMM keeps between SetSrcCoord/SetTrgCoord and repeated Eval calls."""

    def __init__(self, name, dtype=np.float64, devices=(0,), ctx=None):
        self.info = kernel_info(name)
        self.dtype = np.dtype(dtype)
        self.real = _real_of(dtype)
        self.ctx = ctx
        self.Nt = self.Ns = 0
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().sctl_amd_op_create(self.info["id"], self.real, devs, len(devices), C.byref(self._h)), "op_create")

    def set_targets(self, r_trg):
        self.Nt = r_trg.size // 3
        self._dot = False
        _check(lib().sctl_amd_op_set_targets(self._h, self.Nt, _np_ptr(r_trg, self.dtype, self.Nt * 3, "r_trg")), "op_set_targets")

    def set_sources(self, r_src, n_src=None):
        self.Ns = r_src.size // 3
        _check(lib().sctl_amd_op_set_sources(self._h, self.Ns, _np_ptr(r_src, self.dtype, self.Ns * 3, "r_src"),
                                             _np_ptr(n_src, self.dtype, self.Ns * self.info["nd"], "n_src")), "op_set_sources")

    def set_source_weights(self, weights):
        """Quadrature weights applied to every density on the device (None clears); after set_sources."""
        self._weights = weights is not None
        _check(lib().sctl_amd_op_set_source_weights(self._h, None if weights is None else _np_ptr(weights, self.dtype, self.Ns, "weights")), "op_set_source_weights")

    def set_target_normals(self, n_trg):
        """The kernel's output is contracted with these normals on the device, TrgDim -> TrgDim/3 (None clears); after set_targets."""
        _check(lib().sctl_amd_op_set_target_normals(self._h, None if n_trg is None else _np_ptr(n_trg, self.dtype, self.Nt * 3, "n_trg")), "op_set_target_normals")
        self._dot = n_trg is not None

    def eval(self, v_src, v_trg=None, accumulate=False, digits=-1):
        k1 = self.info["k1"] // 3 if getattr(self, "_dot", False) else self.info["k1"]
        if v_trg is None or v_trg.size != self.Nt * k1:
            v_trg = np.zeros(self.Nt * k1, dtype=self.dtype)
        keep, cp, cb = _ctx_blob(self.info, self.ctx)
        _check(lib().sctl_amd_op_eval(self._h, _np_ptr(v_src, self.dtype, self.Ns * self.info["k0"], "v_src"),
                                      _np_ptr(v_trg, self.dtype, self.Nt * k1, "v_trg"), 1 if accumulate else 0, digits, cp, cb), "op_eval")
        return v_trg

    def set_near(self, trg_dim, elem_nds_cnt, near_elem_cnt, K_near, near_scatter_index, near_trg_cnt, near_trg_dsp, K_near_cnt=None):
        """Attach the near-field operator of the same BoundaryIntegralOp (the arrays of NearOp) for the current targets."""
        i8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int64)
        nds, near, kcnt, sc, tc, td = i8(elem_nds_cnt), i8(near_elem_cnt), i8(K_near_cnt), i8(near_scatter_index), i8(near_trg_cnt), i8(near_trg_dsp)
        K = np.ascontiguousarray(K_near, dtype=self.dtype)
        p = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)
        _check(lib().sctl_amd_op_set_near(self._h, self.info["k0"], trg_dim, nds.size, p(nds), p(near), p(kcnt), p(K), p(sc), p(tc), p(td)), "op_set_near")
        self._near_len, self._near_k1 = int(nds.sum()) * self.info["k0"], trg_dim

    def eval_potential(self, v_src_far, f_near, v_trg=None, accumulate=False, digits=-1):
        """Far field + attached near field in one pass over the devices (sctl_amd_op_eval_potential): ComputePotential."""
        k1 = self._near_k1
        if v_trg is None or v_trg.size != self.Nt * k1:
            v_trg = np.zeros(self.Nt * k1, dtype=self.dtype)
        keep, cp, cb = _ctx_blob(self.info, self.ctx)
        _check(lib().sctl_amd_op_eval_potential(self._h, _np_ptr(v_src_far, self.dtype, self.Ns * self.info["k0"], "v_src_far"),
                                                _np_ptr(f_near, self.dtype, self._near_len, "f_near"), _np_ptr(v_trg, self.dtype, self.Nt * k1, "v_trg"),
                                                1 if accumulate else 0, digits, cp, cb), "op_eval_potential")
        return v_trg

    def close(self):
        if self._h:
            lib().sctl_amd_op_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NearOp:
    """BoundaryIntegralOp::ComputeNearInterac with the operator blocks resident on one GPU (sctl_amd_near_*): built from the
    arrays SetupNear leaves behind (boundary_integral.txx:816-1012), applied once per solver iteration."""

    def __init__(self, src_dim, trg_dim, elem_nds_cnt, near_elem_cnt, K_near, near_scatter_index, near_trg_cnt, near_trg_dsp, K_near_cnt=None, device=0):
        i8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int64)
        self._keep = [i8(elem_nds_cnt), i8(near_elem_cnt), i8(K_near_cnt), np.ascontiguousarray(K_near), i8(near_scatter_index), i8(near_trg_cnt), i8(near_trg_dsp)]
        nds, near, kcnt, K, sc, tc, td = self._keep
        self.dtype = K.dtype
        self.real = _real_of(K.dtype)
        self.device = device
        h = C.c_void_p()
        p = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)
        _check(lib().sctl_amd_near_create(self.real, device, nds.size, src_dim, trg_dim, p(nds), p(near), p(kcnt), p(K), tc.size, p(sc), p(tc), p(td), C.byref(h)),
               "near_create")
        self._h = h
        v = [C.c_int64() for _ in range(5)]
        _check(lib().sctl_amd_near_info(self._h, *[C.byref(x) for x in v]), "near_info")
        self.density_len, self.potential_len, self.near_entries, self.operator_bytes, self.workgroups = (x.value for x in v)

    def apply(self, F, U=None):
        """U += near field of density F (numpy arrays); U=None starts from zero."""
        F = np.ascontiguousarray(F, dtype=self.dtype)
        if F.size != self.density_len:
            raise SctlAmdError("density must hold %d values" % self.density_len)
        if U is None:
            U = np.zeros(self.potential_len, dtype=self.dtype)
        if U.size != self.potential_len or U.dtype != self.dtype or not U.flags.c_contiguous:
            raise SctlAmdError("potential must be a contiguous %s array of %d values" % (self.dtype, self.potential_len))
        p = lambda a: None if a.size == 0 else a.ctypes.data_as(C.c_void_p)
        _check(lib().sctl_amd_near_apply_host(self._h, p(F), p(U)), "near_apply_host")
        return U

    def apply_device(self, F, U, stream=None):
        """The same on torch CUDA tensors, enqueued on `stream` (default: torch's current stream); U is accumulated into."""
        import torch
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        with torch.cuda.device(F.device):
            st = stream if stream is not None else torch.cuda.current_stream()
            _check(lib().sctl_amd_near_apply_device(self._h, _t_ptr(F, tdt, self.density_len, "F"), _t_ptr(U, tdt, self.potential_len, "U"),
                                                    C.c_void_p(st.cuda_stream)), "near_apply_device")
        return U

    def close(self):
        if getattr(self, "_h", None):
            lib().sctl_amd_near_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ListsPlan:
    """Many (target range x source range) direct sums in one launch (sctl_amd_lists_*): the P2P / U-list shape of a tree code
    (fmm-wrapper.txx:756-786).  Offsets and counts are in points; target ranges must be identical or disjoint."""

    def __init__(self, name, dtype, trg_off, trg_cnt, src_off, src_cnt, Nt, Ns, device=0, ctx=None):
        self.info = kernel_info(name)
        self.dtype = np.dtype(dtype)
        self.real = _real_of(dtype)
        self.ctx, self.Nt, self.Ns, self.device = ctx, int(Nt), int(Ns), device
        arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (trg_off, trg_cnt, src_off, src_cnt)]
        if len({a.size for a in arrs}) != 1:
            raise SctlAmdError("the four list arrays must have one entry per list")
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        self._h = C.c_void_p()
        _check(lib().sctl_amd_lists_create(self.info["id"], self.real, device, arrs[0].size, p(arrs[0]), p(arrs[1]), p(arrs[2]), p(arrs[3]), self.Nt, self.Ns,
                                           C.byref(self._h)), "lists_create")
        v = [C.c_int64() for _ in range(3)]
        _check(lib().sctl_amd_lists_info(self._h, C.byref(v[0]), C.byref(v[1]), C.byref(v[2])), "lists_info")
        self.pairs, self.work_items, self.source_ranges = v[0].value, v[1].value, v[2].value

    def eval_host(self, r_trg, r_src, n_src, v_src, v_trg=None, digits=-1):
        """numpy arrays; v_trg of the right size is accumulated into, otherwise a fresh zeroed result is returned."""
        dt, i = self.dtype, self.info
        if v_trg is None or v_trg.size != self.Nt * i["k1"]:
            v_trg = np.zeros(self.Nt * i["k1"], dtype=dt)
        keep, cp, cb = _ctx_blob(i, self.ctx)
        _check(lib().sctl_amd_lists_eval_host(self._h, _np_ptr(r_trg, dt, self.Nt * 3, "r_trg"), _np_ptr(r_src, dt, self.Ns * 3, "r_src"),
                                              _np_ptr(n_src, dt, self.Ns * i["nd"], "n_src"), _np_ptr(v_src, dt, self.Ns * i["k0"], "v_src"),
                                              _np_ptr(v_trg, dt, self.Nt * i["k1"], "v_trg"), digits, cp, cb), "lists_eval_host")
        return v_trg

    def eval_device(self, r_trg, r_src, n_src, v_src, v_trg=None, digits=-1, stream=None):
        """torch CUDA tensors on the plan's device; enqueued on `stream` (default: torch's current stream); v_trg is accumulated into."""
        import torch
        i = self.info
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        if v_trg is None or v_trg.numel() != self.Nt * i["k1"]:
            v_trg = torch.zeros(self.Nt * i["k1"], dtype=tdt, device=r_trg.device)
        keep, cp, cb = _ctx_blob(i, self.ctx)
        with torch.cuda.device(r_trg.device):
            st = stream if stream is not None else torch.cuda.current_stream()
            _check(lib().sctl_amd_lists_eval_device(self._h, _t_ptr(r_trg, tdt, self.Nt * 3, "r_trg"), _t_ptr(r_src, tdt, self.Ns * 3, "r_src"),
                                                    _t_ptr(n_src, tdt, self.Ns * i["nd"], "n_src"), _t_ptr(v_src, tdt, self.Ns * i["k0"], "v_src"),
                                                    _t_ptr(v_trg, tdt, self.Nt * i["k1"], "v_trg"), digits, cp, cb, C.c_void_p(st.cuda_stream)), "lists_eval_device")
        return v_trg

    def close(self):
        if getattr(self, "_h", None):
            lib().sctl_amd_lists_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def eval_lists_host(name, trg_off, trg_cnt, src_off, src_cnt, r_trg, r_src, n_src, v_src, v_trg=None, digits=-1, ctx=None, device=0):
    """One-shot sctl_amd_eval_lists_host."""
    plan = ListsPlan(name, r_trg.dtype, trg_off, trg_cnt, src_off, src_cnt, r_trg.size // 3, r_src.size // 3, device=device, ctx=ctx)
    try:
        return plan.eval_host(r_trg, r_src, n_src, v_src, v_trg, digits)
    finally:
        plan.close()
