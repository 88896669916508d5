"""The one-process-per-GPU drivers with the HIP evaluator underneath, rehearsed on ONE GPU: two ranks share device 0 and
talk over gloo (RCCL refuses two ranks on one device; the driver's real multi-GPU runs use backend "nccl").  What this covers
that the CPU gloo tests cannot: Morton-ordered slabs through sctl_amd_eval_device_slab (tile-centred path on a slab), the
gather back into the caller's order on device tensors, and the source ring with device buffers."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import sctl_amd
    from sctl_amd.distributed import RingDirectSum, ShardedDirectSum
    O = oracle.restatement()
    ok = []
    g = torch.Generator(device="cuda").manual_seed(7)
    nt, ns = (1 << 18) + 3, 70001
    xt = torch.rand(nt * 3, dtype=torch.float64, device="cuda", generator=g)
    xs = torch.rand(ns * 3, dtype=torch.float64, device="cuda", generator=g)
    f = torch.rand(ns, dtype=torch.float64, device="cuda", generator=g) - 0.5
    single = sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f)
    for compact in (True, False):
        op = ShardedDirectSum("Laplace3D-FxU", compact=compact)
        u = op.eval(xt, xs, None, f)
        u2 = op.eval(xt, xs, None, f, out=u)                      # overwrite semantics, cached permutation
        ok.append(u.numel() == nt and float((u - single).norm() / single.norm()) <= 2e-14 and torch.equal(u, u2))
    sel = np.arange(0, nt, nt // 128)
    ref = O.eval("Laplace3D-FxU", xt.cpu().numpy().reshape(nt, 3)[sel].ravel().copy(), xs.cpu().numpy(), None, f.cpu().numpy())
    ok.append(np.linalg.norm(u.cpu().numpy()[sel] - ref) <= 1e-12 * np.linalg.norm(ref))
    # the slab of a rank went down the tile-centred path only with the Morton partition
    n_loc = nt // world
    ok.append(sctl_amd.plan("Laplace3D-FxU", 0, n_loc, ns, nt_whole=nt)["path"] == "tile-centred" and sctl_amd.plan("Laplace3D-FxU", 0, n_loc, ns)["path"] == "exact")
    # ring: every rank owns a part of the targets and of the sources (Stokes double layer: normals travel too)
    info = sctl_amd.kernel_info("Stokes3D-DxU")
    nts, nss = [3001, 1999], [2500, 4100]
    gen = torch.Generator(device="cuda").manual_seed(11)
    parts = [[torch.rand(n * 3, dtype=torch.float64, device="cuda", generator=gen) for n in nts],
             [torch.rand(n * 3, dtype=torch.float64, device="cuda", generator=gen) for n in nss],
             [torch.rand(n * 3, dtype=torch.float64, device="cuda", generator=gen) - 0.5 for n in nss],
             [torch.rand(n * info["k0"], dtype=torch.float64, device="cuda", generator=gen) - 0.5 for n in nss]]
    ur = RingDirectSum("Stokes3D-DxU").eval(parts[0][rank], parts[1][rank], parts[2][rank], parts[3][rank])
    refr = O.eval("Stokes3D-DxU", parts[0][rank].cpu().numpy(), *[torch.cat(p).cpu().numpy() for p in parts[1:]])
    ok.append(np.linalg.norm(ur.cpu().numpy() - refr) <= 1e-12 * np.linalg.norm(refr))
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as fh:
        fh.write(" ".join("ok" if x else "BAD" for x in ok))
    dist.destroy_process_group()


def test_sharded_and_ring_drivers_on_the_hip_path(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = open(os.path.join(str(tmp_path), "rank%d" % r)).read()
        assert res and "BAD" not in res, (r, res)


def _nccl_worker(rank, world, port, out_dir):
    """One rank per GPU on backend "nccl" (= RCCL over xGMI): the path the driver's multi-GPU bench takes."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import sctl_amd
    from sctl_amd.distributed import ShardedDirectSum
    g = torch.Generator(device="cuda").manual_seed(7)
    nt, ns = (1 << 17) + 5, 30011                                  # ragged: exercises the padded all-gather too
    xt = torch.rand(nt * 3, dtype=torch.float64, device="cuda", generator=g)
    xs = torch.rand(ns * 3, dtype=torch.float64, device="cuda", generator=g)
    f = torch.rand(ns, dtype=torch.float64, device="cuda", generator=g) - 0.5
    single = sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f)
    ok = []
    for compact in (True, False):
        u = ShardedDirectSum("Laplace3D-FxU", compact=compact).eval(xt, xs, None, f)
        ok.append(float((u - single).norm() / single.norm()) <= 2e-14)
    nt2 = world * 4096                                             # equal slabs: the fused all_gather_into_tensor
    u = ShardedDirectSum("Laplace3D-FxU").eval(xt[:nt2 * 3], xs, None, f)
    ok.append(float((u - single[:nt2]).norm() / single[:nt2].norm()) <= 2e-14)
    # the three collectives bench.py issues, called directly (with one rank they still go through RCCL)
    mine = torch.full((1024,), float(rank + 1), dtype=torch.float64, device="cuda")
    every = torch.empty(1024 * world, dtype=torch.float64, device="cuda")
    dist.all_gather_into_tensor(every, mine)
    ok.append(bool((every.view(world, 1024)[:, 0] == torch.arange(1, world + 1, dtype=torch.float64, device="cuda")).all()))
    t = torch.tensor([float(rank)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    ok.append(float(t.item()) == world - 1)
    with open(os.path.join(out_dir, "nccl%d" % rank), "w") as fh:
        fh.write(" ".join("ok" if x else "BAD" for x in ok))
    dist.destroy_process_group()


def test_sharded_direct_sum_over_rccl(tmp_path):
    """Backend "nccl" with one rank per visible GPU (up to 4).  On a one-GPU box that is a single rank: the slab driver is then
    trivial, but the process group, the all-gather, the all-reduce and the barrier still run on RCCL."""
    import torch
    import torch.multiprocessing as mp
    world = max(1, min(torch.cuda.device_count(), 4))
    mp.spawn(_nccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = open(os.path.join(str(tmp_path), "nccl%d" % r)).read()
        assert res and "BAD" not in res, (r, res)


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_end_to_end():
    """`python bench.py --gpus 2` exactly as the driver may invoke it (no torchrun, no rank environment).  On a one-GPU box the two
    ranks share device 0 over gloo (SCTL_AMD_BENCH_REHEARSAL=1); with two or more GPUs this is the real RCCL run."""
    import torch
    rehearsal = torch.cuda.device_count() < 2
    line = _run_bench({"SCTL_AMD_BENCH_REHEARSAL": "1"} if rehearsal else {}, "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["value"] > 1e11
    assert line["config"]["n_trg"] == 1 << 20 and line["dtype"] == "f64" and "roofline" in line


@pytest.mark.parametrize("workload,kernel,N,dtype", [("helmholtz", "Helmholtz3D-FxU", 1 << 20, "f64"),        # BASELINE configs[4]: 4 GPUs
                                                     ("laplace_sl_f32", "Laplace3D-FxU", 1 << 23, "f32")])    # configs[3]: the 8-GPU one, 4 ranks here
def test_bench_multi_rank_record_of_the_other_split_configs(workload, kernel, N, dtype):
    """BASELINE's two multi-GPU configs through `python bench.py --gpus 4 --workload ...` at their FULL sizes, and the fields that make an
    N > 1 record prove itself: the backend and world size AS THE PROCESS GROUP REPORTS THEM, one entry per rank with its device and its own
    kernel / collective times, the collective leg timed by its own event pair, and n1_equiv_ms (slowest rank's kernel time x N).
    A one-GPU box allows 6 processes on the card (this one included), so config 4's eight ranks are rehearsed as four here (its eight-way
    slab arithmetic: tests/test_distributed_cpu.py, world 8); with >= 4 GPUs this is a real RCCL run."""
    import torch
    rehearsal = torch.cuda.device_count() < 4
    line = _run_bench({"SCTL_AMD_BENCH_REHEARSAL": "1"} if rehearsal else {}, "--gpus", "4", "--steps", "1", "--warmup", "0", "--workload", workload)
    assert line["n_gpus"] == 4 and line["config"]["kernel"] == kernel and line["config"]["n_trg"] == N and line["dtype"] == dtype
    col = line["collective"]
    assert col["world_size"] == 4 and col["backend"] == ("gloo" if rehearsal else "nccl") and col["rehearsal_shared_gpu"] == rehearsal
    assert col["distinct_devices"] == (1 if rehearsal else 4)
    ranks = col["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and sum(r["slab_targets"] for r in ranks) == N
    assert all(r["kernel_ms"] > 0 and r["gather_ms"] > 0 for r in ranks)
    assert line["n1_equiv_ms"] == pytest.approx(4 * max(r["kernel_ms"] for r in ranks))
    assert "cpu_baseline" not in line and line["roofline"]["kernel_ms"] == pytest.approx(ranks[0]["kernel_ms"])
    if rehearsal:      # the ranks share one GPU: their kernels run one after the other, so a step takes about the sum of them
        assert line["ms_per_step"] >= 0.6 * sum(r["kernel_ms"] for r in ranks) / 4


def _split_worker(rank, world, port, out_dir):
    """The two other split configs at reduced size on the shared GPU: sharded result == single-GPU result (Helmholtz with its
    wavenumber context; fp32 Laplace through the tile-centred path on Morton slabs)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sctl_amd
    from sctl_amd.distributed import ShardedDirectSum
    ok = []
    g = torch.Generator(device="cuda").manual_seed(3)
    nt, ns = (1 << 16) + 7, 20011
    xt = torch.rand(nt * 3, dtype=torch.float64, device="cuda", generator=g)
    xs = torch.rand(ns * 3, dtype=torch.float64, device="cuda", generator=g)
    f = torch.rand(ns * 2, dtype=torch.float64, device="cuda", generator=g) - 0.5
    k = np.array([7.5, 0.3])
    single = sctl_amd.eval_device("Helmholtz3D-FxU", xt, xs, None, f, ctx=k)
    u = ShardedDirectSum("Helmholtz3D-FxU", ctx=k).eval(xt, xs, None, f)
    ok.append(float((u - single).norm() / single.norm()) <= 1e-14)
    nt, ns = (1 << 20) + 5, 1 << 16
    xt = torch.rand(nt * 3, dtype=torch.float32, device="cuda", generator=g)
    xs = torch.rand(ns * 3, dtype=torch.float32, device="cuda", generator=g)
    f = torch.rand(ns, dtype=torch.float32, device="cuda", generator=g) - 0.5
    single = sctl_amd.eval_device("Laplace3D-FxU", xt, xs, None, f)
    u = ShardedDirectSum("Laplace3D-FxU").eval(xt, xs, None, f)
    ok.append(float((u - single).norm() / single.norm()) <= 2e-6)        # fp32: tile boundaries differ between the slab and the whole set
    ok.append(sctl_amd.plan("Laplace3D-FxU", 1, nt // world, ns, nt_whole=nt)["path"] == "tile-centred")
    with open(os.path.join(out_dir, "split%d" % rank), "w") as fh:
        fh.write(" ".join("ok" if x else "BAD" for x in ok))
    dist.destroy_process_group()


def test_split_of_the_helmholtz_and_fp32_configs_equals_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    world = 4
    mp.spawn(_split_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = open(os.path.join(str(tmp_path), "split%d" % r)).read()
        assert res and "BAD" not in res, (r, res)
