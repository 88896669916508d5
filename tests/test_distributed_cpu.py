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


def _worker(rank, world, port, Nt, Ns, name, out_dir):
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

    op = ShardedDirectSum(name, local_eval=oracle_eval)
    t = [torch.from_numpy(a) for a in (xt, xs, xn, f)]
    u = op.eval(*t)
    u2 = op.eval(*t, out=u)                        # EvalDirect overwrites: a second call gives the same answer
    ref = O.eval(name, xt, xs, xn, f, nthreads=2)
    t0, t1 = slab_bounds(Nt, rank, world)
    ok = (np.linalg.norm(u.numpy() - ref) <= 1e-14 * np.linalg.norm(ref)) and torch.equal(u, u2) and (t1 - t0) in (Nt // world, Nt // world + 1)
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as fh:
        fh.write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,Nt", [(2, 1000), (2, 1001), (3, 1000)])
def test_sharded_direct_sum_gloo(tmp_path, world, Nt):
    mp.spawn(_worker, args=(world, _free_port(), Nt, 300, "Stokes3D-DxU", str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_slab_bounds_cover_all_targets():
    from sctl_amd.distributed import slab_bounds
    for Nt in (0, 1, 7, 1000, 1 << 20):
        for G in (1, 2, 3, 8):
            b = [slab_bounds(Nt, g, G) for g in range(G)]
            assert b[0][0] == 0 and b[-1][1] == Nt and all(b[i][1] == b[i + 1][0] for i in range(G - 1))
