"""Rank-parallel runs from the C++ host surface (SURVEY.md §8f row 4 / VERDICT r1 "missing 3"): include/sctl_amd/comm.hpp's Comm::World()
(rank environment -> TCP rendezvous -> RCCL when every rank has its own GPU) and ParticleFMM::EvalDirect with every rank owning a share
of the targets and of the sources (the contract of fmm-wrapper.txx:504-561).  One process per rank, started here the way mpirun /
torchrun would (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  CPU: the communicator alone (uneven all-gather, barrier) with three
ranks.  GPU: two and three ranks sharing device 0 (data over the rendezvous sockets: RCCL refuses ranks that share a GPU) and, where
the box has at least two GPUs, one GPU per rank over RCCL; every rank's slice of the potential against the oracle on the global problem."""
import os
import socket
import subprocess

import numpy as np
import pytest

from conftest import rel_l2
from sctl_amd.rand48 import Rand48
from test_cpp_host import _build, _read_vector


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(exe, world, args, extra_env=None, timeout=300, master_addr="127.0.0.1"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR=master_addr, MASTER_PORT=str(port), **(extra_env(r) if extra_env else {}))
        procs.append(subprocess.Popen([exe] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def test_comm_world_host_collectives_three_ranks(tmp_path):
    exe = _build(tmp_path, "fmm_dist")
    outs = _launch(exe, 3, ["10", str(tmp_path / "x"), "hostonly"], timeout=120)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, e)
        assert "rank %d of 3" % r in o and "host collectives ok" in o and "transport sockets" in o


def test_comm_world_resolves_a_host_name(tmp_path):
    """MASTER_ADDR as launchers export it — a host name, not only dotted IPv4 (getaddrinfo in sctl_amd_comm_create)."""
    exe = _build(tmp_path, "fmm_dist")
    outs = _launch(exe, 2, ["10", str(tmp_path / "x"), "hostonly"], timeout=120, master_addr="localhost")
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, e)
        assert "rank %d of 2" % r in o and "host collectives ok" in o
    outs = _launch(exe, 2, ["10", str(tmp_path / "x"), "hostonly"], timeout=120, master_addr="no-such-host.invalid")
    assert all(rc != 0 and "cannot resolve the rendezvous address" in e for rc, o, e in outs), outs


def test_comm_world_is_self_for_independent_tasks(tmp_path):
    """A launcher's task count alone (independent single-rank tasks under srun / mpirun) must not start a rendezvous: World() goes
    rank-parallel only with SCTL_AMD_WORLD_SIZE or with MASTER_ADDR and MASTER_PORT named; SCTL_AMD_COMM=0 switches it off."""
    exe = _build(tmp_path, "fmm_dist")
    drop = ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK", "SCTL_AMD_WORLD_SIZE", "SCTL_AMD_RANK")
    base = {k: v for k, v in os.environ.items() if k not in drop}
    for extra in ({"SLURM_NTASKS": "4", "SLURM_PROCID": "2"}, {"PMI_SIZE": "8", "PMI_RANK": "5"},
                  {"WORLD_SIZE": "2", "RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "SCTL_AMD_COMM": "0"}):
        r = subprocess.run([exe, "10", str(tmp_path / "x"), "hostonly"], env=dict(base, **extra), capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "rank 0 of 1" in r.stdout, (extra, r.stdout, r.stderr)
        # several tasks reported and no rendezvous named: each evaluates on its own sources — said once on stderr (under the reference's contract, the
        # potential from ALL ranks' sources, that is another answer); switched off explicitly it is the user's choice and silent
        warned = "tasks detected" in r.stderr and "NOT rank-parallel" in r.stderr
        assert warned == ("SCTL_AMD_COMM" not in extra), (extra, r.stderr)
        quiet = subprocess.run([exe, "10", str(tmp_path / "x"), "hostonly"], env=dict(base, SCTL_AMD_COMM_QUIET="1", **extra), capture_output=True, text=True, timeout=60)
        assert quiet.returncode == 0 and "tasks detected" not in quiet.stderr


def test_rendezvous_drops_stray_connections_and_keeps_listening():
    """Rank 0's accept loop: a connection that sends something else than this job's hello, and one that says nothing, are closed and the
    rendezvous goes on; then the host all-gather works.  (Two ranks as two threads of this process; ctypes releases the GIL in the calls.)"""
    import ctypes as C
    import threading
    import time
    import sctl_amd
    L = sctl_amd.lib()
    L.sctl_amd_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.sctl_amd_comm_allgatherv_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.sctl_amd_comm_destroy.argtypes = [C.c_void_p]
    L.sctl_amd_comm_destroy.restype = None
    port = _free_port()
    handles, rcs, got = [C.c_void_p(), C.c_void_p()], [None, None], [None, None]

    def rank(r):
        rcs[r] = L.sctl_amd_comm_create(r, 2, b"localhost", port, -1, 1, C.byref(handles[r]))          # flags = sockets only
        if rcs[r] == 0:
            mine = np.full(r + 2, float(r + 1))
            out = np.zeros(5)
            sizes = (C.c_int64 * 2)()
            rcs[r] = L.sctl_amd_comm_allgatherv_host(handles[r], mine.ctypes.data, mine.nbytes, out.ctypes.data, out.nbytes, sizes)
            got[r] = (out, list(sizes))

    t0 = threading.Thread(target=rank, args=(0,))
    t0.start()
    strays = []
    for payload in (b"GET / HTTP/1.0\r\n\r\n" + b"x" * 32, None):            # a foreign protocol; a silent connection (dropped after its 10 s timeout)
        for _ in range(100):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=1)
                break
            except OSError:
                time.sleep(0.05)
        if payload:
            s.sendall(payload)
        strays.append(s)
    t1 = threading.Thread(target=rank, args=(1,))
    t1.start()
    t0.join(60)
    t1.join(60)
    assert not t0.is_alive() and not t1.is_alive()
    assert rcs == [0, 0], (rcs, sctl_amd.last_error())
    for r in range(2):
        assert got[r][1] == [16, 24] and np.array_equal(got[r][0], [1, 1, 2, 2, 2])
    for s in strays:
        s.close()
    for h in handles:
        L.sctl_amd_comm_destroy(h)


def _cut(N, r, np_):
    return N * (r * (r + 1) // 2) // (np_ * (np_ + 1) // 2)


def _check_slices(O, tmp_path, N, world):
    g = Rand48(0)
    xt, xd, nd, fd, xs, fs = (g.drand48(N * 3) - 0.5 for _ in range(6))
    ref = (O.eval("Stokes3D-DxU", xt, xd, nd, fd) + O.eval("Stokes3D-FxU", xt, xs, None, fs)).reshape(N, 3)
    # after the LAST rank flipped the normals of its double-layer sources (it owns the first slice: the shares run the other way round)
    nd2 = nd.copy().reshape(N, 3)
    nd2[:_cut(N, 1, world)] *= -1
    ref2 = (O.eval("Stokes3D-DxU", xt, xd, nd2.ravel(), fd) + O.eval("Stokes3D-FxU", xt, xs, None, fs)).reshape(N, 3)
    for r in range(world):
        t0, t1 = _cut(N, r, world), _cut(N, r + 1, world)
        for tag, want in ((".r%d" % r, ref), (".moved.r%d" % r, ref2)):
            u = _read_vector(str(tmp_path / "u") + tag)
            assert u.size == (t1 - t0) * 3
            assert rel_l2(u, want[t0:t1]) <= 1e-12, (r, tag, rel_l2(u, want[t0:t1]))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_rank_parallel_eval_direct_ranks_sharing_one_gpu(tmp_path, O, world):
    exe = _build(tmp_path, "fmm_dist")
    N = 6000
    outs = _launch(exe, world, [str(N), str(tmp_path / "u")], extra_env=lambda r: {"LOCAL_RANK": "0"})
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, o, e)
        assert "transport sockets" in o
    _check_slices(O, tmp_path, N, world)


@pytest.mark.gpu
def test_rank_parallel_eval_direct_over_rccl(tmp_path, O):
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has %d" % ngpu)
    world = min(ngpu, 4)
    exe = _build(tmp_path, "fmm_dist")
    N = 20000
    outs = _launch(exe, world, [str(N), str(tmp_path / "u")], extra_env=lambda r: {"LOCAL_RANK": str(r), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, o, e)
        assert "transport rccl" in o
    _check_slices(O, tmp_path, N, world)


@pytest.mark.gpu
def test_rccl_binding_on_one_gpu():
    """The dlopen()ed RCCL entry points (ncclGetUniqueId, ncclCommInitRank with the id passed by value, grouped ncclSend / ncclRecv, destroy)
    on a one-rank communicator: what the one-GPU box can check of the RCCL data path of sctl_amd/csrc/comm.hip."""
    import ctypes as C
    import sctl_amd
    L = sctl_amd.lib()
    L.sctl_amd_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.sctl_amd_comm_selftest.argtypes = [C.c_void_p, C.c_int64]
    L.sctl_amd_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.sctl_amd_comm_destroy.argtypes = [C.c_void_p]
    L.sctl_amd_comm_destroy.restype = None
    h = C.c_void_p()
    assert L.sctl_amd_comm_create(0, 1, None, 0, 0, 2, C.byref(h)) == 0, sctl_amd.last_error()      # flags = SCTL_AMD_COMM_FORCE_RCCL
    transport = C.c_int(-1)
    assert L.sctl_amd_comm_info(h, None, None, None, C.byref(transport)) == 0 and transport.value == 1
    assert L.sctl_amd_comm_selftest(h, 1 << 20) == 0, sctl_amd.last_error()
    L.sctl_amd_comm_destroy(h)
    plain = C.c_void_p()
    assert L.sctl_amd_comm_create(0, 1, None, 0, 0, 0, C.byref(plain)) == 0
    assert L.sctl_amd_comm_selftest(plain, 64) == -2                                              # sockets-only communicator
    L.sctl_amd_comm_destroy(plain)
