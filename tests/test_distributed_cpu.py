"""The N>1 path (target slabs + one all-gather) under gloo with world_size 2 and 3 on CPU.  The local evaluator is the
CPU oracle here — injected by the test, never by product code — so what is tested is the partition formula
(fmm-wrapper.txx:507), ragged slabs and the gather, against a single-rank evaluation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Nt, Ns, name, out_dir, compact=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from sctl_amd.distributed import ShardedDirectSum, slab_bounds
    O = oracle.restatement()
    info = O.info(name)
    rng = np.random.default_rng(4)
    xt, xs = rng.random(Nt * 3), rng.random(Ns * 3)
    xn, f = rng.random(Ns * info["nd"]) - 0.5, rng.random(Ns * info["k0"]) - 0.5

    def oracle_eval(r_trg_slab, r_src, n_src, v_src, v_out):
        v = O.eval(name, r_trg_slab.numpy().copy(), r_src.numpy(), n_src.numpy(), v_src.numpy(), nthreads=2)
        v_out += torch.from_numpy(v)
        return v_out

    op = ShardedDirectSum(name, local_eval=oracle_eval, compact=compact)
    t = [torch.from_numpy(a) for a in (xt, xs, xn, f)]
    u = op.eval(*t)
    u2 = op.eval(*t, out=u)                        # EvalDirect overwrites: a second call gives the same answer
    ref = O.eval(name, xt, xs, xn, f, nthreads=2)
    t0, t1 = slab_bounds(Nt, rank, world)
    idx = op.slab_indices(t[0]).numpy()               # which targets this rank evaluated: a compact box, or an index range
    box = (xt.reshape(-1, 3)[idx].max(0) - xt.reshape(-1, 3)[idx].min(0)).prod()
    ok = (np.linalg.norm(u.numpy() - ref) <= 1e-14 * np.linalg.norm(ref)) and torch.equal(u, u2) and (t1 - t0) in (Nt // world, Nt // world + 1)
    ok = ok and (t0, t1) == ((Nt * rank) // world, (Nt * (rank + 1)) // world)      # the reference's rank formula, fmm-wrapper.txx:507
    ok = ok and idx.size == t1 - t0 and ((box < 0.75 or world != 2) if compact else np.array_equal(idx, np.arange(t0, t1)))
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as fh:
        fh.write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,Nt,compact", [(2, 1000, True), (2, 1001, True), (3, 1000, True), (2, 1001, False), (8, 1003, True), (8, 1003, False)])
def test_sharded_direct_sum_gloo(tmp_path, world, Nt, compact):
    """Slabs cut from the Morton order (default) or by index: the same potential in the caller's order either way."""
    mp.spawn(_worker, args=(world, _free_port(), Nt, 300, "Stokes3D-DxU", str(tmp_path), compact), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_slab_bounds_cover_all_targets():
    from sctl_amd.distributed import slab_bounds
    for Nt in (0, 1, 7, 1000, 1 << 20):
        for G in (1, 2, 3, 8):
            b = [slab_bounds(Nt, g, G) for g in range(G)]
            assert b[0][0] == 0 and b[-1][1] == Nt and all(b[i][1] == b[i + 1][0] for i in range(G - 1))


def _ring_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from sctl_amd.distributed import RingDirectSum
    O = oracle.restatement()
    info = O.info(name)
    rng = np.random.default_rng(8)
    nts = [211, 0, 57][:world] if world == 3 else [300, 123]        # unequal blocks, one rank without targets
    nss = [150, 333, 1][:world] if world == 3 else [200, 77]
    xt_all = [rng.random(n * 3) for n in nts]
    xs_all = [rng.random(n * 3) for n in nss]
    xn_all = [rng.random(n * info["nd"]) - 0.5 for n in nss]
    f_all = [rng.random(n * info["k0"]) - 0.5 for n in nss]

    def oracle_eval(r_trg, r_src, n_src, v_src, v_out):
        v = O.eval(name, r_trg.numpy().copy(), r_src.numpy().copy(), n_src.numpy().copy(), v_src.numpy().copy(), nthreads=2)
        v_out += torch.from_numpy(v)
        return v_out

    op = RingDirectSum(name, local_eval=oracle_eval)
    u = op.eval(*[torch.from_numpy(a[rank]) for a in (xt_all, xs_all, xn_all, f_all)])
    ref = O.eval(name, xt_all[rank], np.concatenate(xs_all), np.concatenate(xn_all), np.concatenate(f_all), nthreads=2) if nts[rank] else np.zeros(0)
    ok = u.numel() == nts[rank] * info["k1"] and (nts[rank] == 0 or np.linalg.norm(u.numpy() - ref) <= 1e-13 * np.linalg.norm(ref))
    with open(os.path.join(out_dir, "ring%d" % rank), "w") as fh:
        fh.write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_direct_sum_gloo(tmp_path, world):
    """Partitioned inputs, sources rotating around the ring (the reference's MPI EvalDirect, fmm-wrapper.txx:537-558)."""
    mp.spawn(_ring_worker, args=(world, _free_port(), "Laplace3D-DxU", str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "ring%d" % r)).read() == "ok"


def test_morton_order_is_a_permutation_and_local():
    from sctl_amd.distributed import morton_order
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3 * 4096, dtype=torch.float64, generator=g) * 5 - 2          # any box, not just the unit cube
    p = morton_order(x)
    assert torch.equal(torch.sort(p).values, torch.arange(4096))
    assert torch.equal(p, morton_order(x.clone()))                                # a pure function of the coordinates
    xs = x.view(-1, 3)[p]
    hop = lambda pts: (pts[1:] - pts[:-1]).norm(dim=1).mean()                     # neighbours on the curve are neighbours in space
    assert hop(xs) < 0.2 * hop(x.view(-1, 3))
    assert morton_order(torch.zeros(0, dtype=torch.float64)).numel() == 0
    same = torch.ones(3 * 10, dtype=torch.float32)                               # degenerate box: stable order
    assert torch.equal(morton_order(same), torch.arange(10))


def _realloc_worker(rank, world, port, out_dir):
    """Two target clouds of the SAME size, the first freed before the second is allocated (the caching allocator — or malloc —
    may hand out the same address): the cached Morton order must not survive from one to the other."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from sctl_amd.distributed import ShardedDirectSum
    O = oracle.restatement()
    name, Nt, Ns = "Laplace3D-FxU", 700, 200
    rng = np.random.default_rng(21)
    xs, f = rng.random(Ns * 3), rng.random(Ns) - 0.5

    def oracle_eval(r_trg_slab, r_src, n_src, v_src, v_out):
        v_out += torch.from_numpy(O.eval(name, r_trg_slab.numpy().copy(), r_src.numpy(), None, v_src.numpy(), nthreads=2))
        return v_out

    op = ShardedDirectSum(name, local_eval=oracle_eval)
    ok, ptrs = True, []

    def once(seed):
        xt = torch.from_numpy(np.random.default_rng(seed).random(Nt * 3))      # a fresh tensor, version 0, dropped on return
        ptrs.append(xt.data_ptr())
        u = op.eval(xt, torch.from_numpy(xs), None, torch.from_numpy(f))
        ref = O.eval(name, xt.numpy(), xs, None, f, nthreads=2)
        return np.linalg.norm(u.numpy() - ref) <= 1e-14 * np.linalg.norm(ref)

    for seed in (1, 2, 3, 4):
        op.invalidate() if seed == 4 else None
        ok = ok and once(seed)
    # in-place change of a tensor the operator has seen: the version counter invalidates the cache
    xt = torch.from_numpy(rng.random(Nt * 3))
    op.eval(xt, torch.from_numpy(xs), None, torch.from_numpy(f))
    xt.mul_(0.5)
    u = op.eval(xt, torch.from_numpy(xs), None, torch.from_numpy(f))
    ref = O.eval(name, xt.numpy(), xs, None, f, nthreads=2)
    ok = ok and np.linalg.norm(u.numpy() - ref) <= 1e-14 * np.linalg.norm(ref)
    with open(os.path.join(out_dir, "realloc%d" % rank), "w") as fh:
        fh.write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_cached_target_order_is_tied_to_the_tensor_not_its_address(tmp_path):
    mp.spawn(_realloc_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(os.path.join(str(tmp_path), "realloc%d" % r)).read() == "ok"


def test_bench_starts_its_own_ranks_when_asked_for_several_gpus():
    """`python bench.py --gpus 2` with no rank environment must launch two ranks itself (the driver's N > 1 command line may be
    the plain one).  Without a GPU each rank stops with bench.py's own message — which shows that two ranks were started and
    that the parent hands a failing exit code on; with GPUs the same path is run for real by tests/test_gpu_distributed.py."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]
