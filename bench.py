#!/usr/bin/env python3
"""Headline benchmark: pair-interactions/s of the Laplace single-layer N x N direct sum on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload laplace_sl|laplace_sldl|stokeslet|helmholtz|laplace_sl_f32|stokeslet_f32|laplace_sl_16k|p2p_lists|near_apply]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full evaluation of the hot path over one synthetic point cloud that is already resident in HBM:
zero the potential, run GenericKernel::Eval's device replacement over all pairs, (N > 1) all-gather the slabs.
At N = 1 the workload is BASELINE.json's headline: Laplace3D single layer, 2^20 sources x 2^20 targets, fp64.
For N > 1 the SAME problem is split: targets are block-partitioned over the ranks, sources replicated, one RCCL
all-gather of the potential slabs per step (strong scaling: north_star asks for >= 6x at 8 GPUs on this problem).

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` and, at N = 1, `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kernel, N, dtype, description)
    "laplace_sl": ("Laplace3D-FxU", 1 << 20, "f64", "Laplace3D single layer (Laplace3D-FxU), 2^20 x 2^20, fp64 [BASELINE headline / north_star target]"),
    "laplace_sldl": ("Laplace3D-FDxUdU", 1 << 20, "f64", "Laplace3D SL+DL potential+gradient (Laplace3D-FDxUdU), 2^20 x 2^20, fp64 [BASELINE configs[1]]"),
    "stokeslet": ("Stokes3D-FxU", 1 << 18, "f64", "Stokes3D Stokeslet (Stokes3D-FxU), 2^18 x 2^18, fp64 [BASELINE configs[2]]"),
    "laplace_sl_f32": ("Laplace3D-FxU", 1 << 23, "f32", "Laplace3D single layer, 2^23 x 2^23, fp32 [BASELINE configs[3]; needs >= 8 GPUs to finish in minutes]"),
    "helmholtz": ("Helmholtz3D-FxU", 1 << 20, "f64", "Helmholtz3D single layer k = 7.5 + 0.3i (Helmholtz3D-FxU), 2^20 x 2^20, fp64 [BASELINE configs[4]]"),
    "stokeslet_f32": ("Stokes3D-FxU", 1 << 20, "f32", "Stokes3D Stokeslet (Stokes3D-FxU), 2^20 x 2^20, fp32 [not a BASELINE config: the reference runs every functor at Real = float, generic-kernel.txx:76]"),
    "laplace_sl_16k": ("Laplace3D-FxU", 1 << 14, "f64", "Laplace3D single layer, 2^14 x 2^14, fp64 [BASELINE configs[0], the reference's CPU-runnable case]"),
}
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # vector FMA peaks, BASELINE.md §3 / MI355X_MICROARCH.md chip table


def cpu_impl():
    """(No OMP_PROC_BIND: with "close" the reference ran 38 % SLOWER on the GPU box's 128-thread host — 3.2e10 against 5.2e10 pairs/s — so
    the threads are left to the OS, which is the better number for the CPU side.)
    The CPU side of the `cpu_baseline` legs — the ONLY place bench.py touches oracle/ (a reported comparator, never the thing measured):
    the reference's own OpenMP + Vec<> path compiled from its headers (oracle/_ref, kind "reference"), else the CPU restatement ("port")."""
    import oracle
    return oracle.reference() or oracle.restatement()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs among the CPUs of its affinity mask (SMT siblings count once)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for c in cpus:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            with open(base + "physical_package_id") as fh:
                pkg = fh.read().strip()
            with open(base + "core_id") as fh:
                seen.add((pkg, fh.read().strip()))
        return len(seen) or len(cpus)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(kernel, N, dtype, budget_s=12.0):
    """The reference's own OpenMP + Vec<> path (oracle/_ref, kind "reference") — or the CPU restatement ("port") when the
    compiled reference is not usable on this host — timed on a bounded target subset against ALL sources."""
    impl = cpu_impl()
    info = impl.info(kernel)
    dt = np.float64 if dtype == "f64" else np.float32
    rng = np.random.default_rng(0)
    xs = rng.random(N * 3).astype(dt)
    xn = (rng.random(N * info["nd"]) - 0.5).astype(dt)
    f = (rng.random(N * info["k0"]) - 0.5).astype(dt)
    ctx = np.array([7.5, 0.3]) if kernel.startswith("Helmholtz") else None

    def run(nt):
        xt = rng.random(nt * 3).astype(dt)
        t = time.perf_counter()
        impl.eval(kernel, xt, xs, xn, f, ctx=ctx)
        return time.perf_counter() - t

    nt0 = max(64, min(N, (1 << 32) // N))
    run(nt0)                                  # thread pool / page warm-up
    rate = nt0 * N / run(nt0)
    nt = int(min(N, max(nt0, budget_s * rate / N)))
    nt -= nt % 64
    secs = run(nt)
    return {"value": nt * N / secs, "unit": "pair-interactions/s", "cores": min(physical_cores(), impl.num_threads()), "threads": impl.num_threads(), "kind": impl.kind,
            "isa": getattr(impl, "isa", "x86-64-v3"), "cpu": cpu_model(),
            "sample": "%d targets x %d sources (all sources, target subset; work is linear in targets), %.1f s, %s" % (nt, N, secs, kernel)}


def read_traffic(workload):
    """HBM bytes per launch of the workload's dominant kernel, from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json; counters
    cannot be read from inside this process).  None — never a stale number — unless the kernel the record names still exists in the
    library that is being measured (its name tokens must appear among the code object's symbols)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            rec = json.load(fh).get(workload)
        if not rec or not rec.get("symbol_tokens"):
            return None
        import sctl_amd
        with open(sctl_amd.library_path(), "rb") as fh:
            blob = fh.read()
        return rec["hbm_bytes_per_launch"] if all(t.encode() in blob for t in rec["symbol_tokens"]) else None
    except (OSError, ValueError, KeyError):
        return None


def bench_near(args):
    """Extra workload: one application of a device-resident near-field operator,
    BoundaryIntegralOp::ComputeNearInterac (boundary_integral.txx:1079-1142).  HBM-bound: the algorithmic traffic of a step is
    the bytes of K_near (every operator entry is read once); densities, U_near and the index arrays are < 1 % of that."""
    import torch
    import sctl_amd
    if args.gpus != 1:
        raise SystemExit("bench.py --workload near_apply runs on one GPU (the operator of a rank is applied where its targets live)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: sctl_amd has no CPU path to measure")
    nelem, nds, near, k0, k1 = 2048, 48, 400, 3, 3            # Stokes-like: 144 x 1200 blocks, 2.8 GB of fp64 operator
    rng = np.random.default_rng(0)
    n_near = nelem * near
    ntrg = n_near // 8                                         # every target is near ~8 elements
    K = rng.standard_normal(nelem * nds * k0 * near * k1)
    trg = rng.integers(0, ntrg, n_near)
    order = np.argsort(trg, kind="stable")
    cnt = np.bincount(trg, minlength=ntrg)
    dsp = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    op = sctl_amd.NearOp(k0, k1, np.full(nelem, nds), np.full(nelem, near), K, order, cnt, dsp)
    F = torch.randn(op.density_len, dtype=torch.float64, device="cuda")
    U = torch.zeros(op.potential_len, dtype=torch.float64, device="cuda")
    for _ in range(args.warmup):
        op.apply_device(F, U)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tic = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        op.apply_device(F, U)
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - tic
    k_ms = e0.elapsed_time(e1) / args.steps
    entries = op.operator_bytes / 8
    achieved = op.operator_bytes / (k_ms * 1e-3) / 1e9
    line = {"metric": "operator entries applied/s, BoundaryIntegralOp near field (ComputeNearInterac)", "value": entries / (elapsed / args.steps),
            "unit": "matrix-entries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "near-field operator: %d elements, blocks %d x %d (Stokes-like), %.2f GB of K_near resident in HBM, %d targets" %
                                   (nelem, nds * k0, near * k1, op.operator_bytes / 1e9, ntrg), "workgroups": op.workgroups},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": read_traffic("near_apply"),
                         "kernel_ms": k_ms, "note": "algorithmic bytes = sizeof(K_near): 8 B per operator entry, read once per application"}}
    # Informational, outside the timed region: the whole BoundaryIntegralOp::ComputePotential (boundary_integral.txx:608-614) for this operator
    # from HOST arrays, as a C++ caller has them — far field (Stokeslet over the elements' nodes as far-field quadrature) + this near field.
    # "fused": sctl_amd_op_eval_potential (densities down once, near field added to the far field on the device, potential up once);
    # "two legs": sctl_amd_op_eval, then sctl_amd_near_apply_host (two round trips over PCIe).
    xs = rng.random(nelem * nds * 3)
    xt = rng.random(ntrg * 3)
    wts = rng.random(nelem * nds) * 1e-3
    far = sctl_amd.DirectOp("Stokes3D-FxU", np.float64)
    far.set_targets(xt)
    far.set_sources(xs)
    far.set_source_weights(wts)
    far.set_near(k1, np.full(nelem, nds), np.full(nelem, near), K, order, cnt, dsp)
    Fh = rng.standard_normal(op.density_len)
    Uh = np.zeros(op.potential_len)
    far.eval_potential(Fh, Fh, v_trg=Uh, digits=10)

    def wall(fn, reps=5):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (time.perf_counter() - t0) / reps

    def two_legs():
        far.eval(Fh, v_trg=Uh, digits=10)
        op.apply(Fh, U=Uh)
    ms_fused = wall(lambda: far.eval_potential(Fh, Fh, v_trg=Uh, digits=10))
    ms_two = wall(two_legs)
    line["compute_potential_from_host"] = {"fused_ms": ms_fused, "two_legs_ms": ms_two, "far_pairs": float(ntrg) * nelem * nds,
                                           "note": "far field Stokes3D-FxU %d x %d at 10 digits + this near field, host arrays in, host array out; not part of `value`" % (ntrg, nelem * nds)}
    far.close()
    if not args.no_cpu_baseline:
        sample = 256                                           # elements; numpy's BLAS GEMV per element, as Matrix::GEMM at :1101
        Kb = K[:sample * nds * k0 * near * k1].reshape(sample, nds * k0, near * k1)
        Fh = rng.standard_normal((sample, nds * k0))
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 5.0:
            for e in range(sample):
                Fh[e] @ Kb[e]
            reps += 1
        secs = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": reps * Kb.size / secs, "unit": "matrix-entries/s", "cores": physical_cores(), "threads": os.cpu_count(), "kind": "port",
                                "sample": "%d of %d element blocks, numpy (BLAS) GEMV per block, %d repetitions, %.1f s" % (sample, nelem, reps, secs)}
    print(json.dumps(line), flush=True)


def bench_lists(args):
    """Extra workload (second half): the near-field (U-list) pass of a uniform tree level as ONE launch of the
    batched list kernel (sctl_amd_lists_*; the per-box-pair call shape of fmm-wrapper.txx:756-786).  2^21 uniform points in 16^3 leaf
    boxes (~512 per box), every box against itself and its up to 26 neighbours, targets == sources (the self-interaction a tree
    code has): 9.5e4 lists, 2.6e10 pair interactions per step.  Compute-bound like the all-pairs sum: same flop convention, same peak."""
    import torch
    import sctl_amd
    from sctl_amd.lists import grid_neighbour_lists
    if args.gpus != 1:
        raise SystemExit("bench.py --workload p2p_lists runs on one GPU (a rank evaluates the lists of the boxes it owns)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: sctl_amd has no CPU path to measure")
    kernel, grid, N = "Laplace3D-FxU", 16, 1 << 21
    rng = np.random.default_rng(0)
    x = rng.random((N, 3))
    box = (np.floor(x[:, 0] * grid) * grid + np.floor(x[:, 1] * grid)) * grid + np.floor(x[:, 2] * grid)
    order = np.argsort(box, kind="stable")
    x = x[order].ravel().copy()                                   # particles stored box by box, as a tree keeps them
    counts = np.bincount(box.astype(np.int64), minlength=grid ** 3)
    lists = grid_neighbour_lists(grid, counts, counts)
    f = rng.random(N) - 0.5
    plan = sctl_amd.ListsPlan(kernel, np.float64, *lists, N, N)
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    u = torch.zeros(N, dtype=torch.float64, device="cuda")

    def step():
        u.zero_()
        plan.eval_device(dx, dx, None, df, v_trg=u, digits=args.digits)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    tic = time.perf_counter()
    for e0, e1 in ev:
        u.zero_()
        e0.record()
        plan.eval_device(dx, dx, None, df, v_trg=u, digits=args.digits)
        e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - tic
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    fpp = sctl_amd.flops_per_pair(kernel)
    achieved = plan.pairs * fpp / (k_ms * 1e-3) / 1e12
    line = {"metric": "pair-interactions/s, near-field lists (P2P) of a uniform tree level", "value": plan.pairs / (elapsed / args.steps), "unit": "pair-interactions/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batched list evaluation: %d (target box x source box) lists, 2^21 uniform points in 16^3 boxes (%d..%d per box), targets == sources, %s"
                                   % (lists[0].size, counts.min(), counts.max(), kernel), "kernel": kernel, "lists": int(lists[0].size), "pairs_per_step": plan.pairs,
                       "work_items": plan.work_items, "digits": args.digits},
            "roofline": {"bound": "mfma", "pipe": "fp64 VALU", "achieved": achieved, "peak": PEAK_TFLOPS["f64"], "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS["f64"],
                         "traffic": read_traffic("p2p_lists"), "flops_per_pair": fpp, "kernel_ms": k_ms,
                         "note": "vector-FMA bound: algorithmic flops = listed pairs x (3 + FLOPS() + 2 K0 K1); one launch per step"}}
    if not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        R = cpu_impl()
        nthreads = min(os.cpu_count() or 1, 32)
        nb_sample = grid ** 3                                      # target boxes of the sample: all of them (a few seconds of CPU work)
        to, tc, so, sc = lists
        sel = np.flatnonzero(np.isin(to, np.unique(to)[:nb_sample]))
        uh = np.zeros(N)

        def work(chunk):                                           # a worker owns whole target boxes: no two threads write one range
            for l in chunk:
                t0, t1, s0, s1 = int(to[l]), int(to[l] + tc[l]), int(so[l]), int(so[l] + sc[l])
                kw = dict(omp=False) if R.kind == "reference" else dict(nthreads=1)
                R.eval(kernel, x[t0 * 3:t1 * 3], x[s0 * 3:s1 * 3], None, f[s0:s1], v_trg=uh[t0:t1], **kw)
        boxes = np.array_split(np.unique(to[sel]), nthreads * 4)
        chunks = [sel[np.isin(to[sel], b)] for b in boxes]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(work, chunks))
        secs = time.perf_counter() - t0
        pairs = int((tc[sel] * sc[sel]).sum())
        line["cpu_baseline"] = {"value": pairs / secs, "unit": "pair-interactions/s", "cores": min(physical_cores(), nthreads), "threads": nthreads, "kind": R.kind,
                                "sample": "the %d lists of the first %d target boxes, one GenericKernel::Eval per list from %d worker threads (as PVFMM calls it), %.1f s"
                                          % (sel.size, nb_sample, nthreads, secs)}
    print(json.dumps(line), flush=True)


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves, one process per GPU, the way the
    driver's own multi-GPU command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...).
    Decided BEFORE anything touches the GPU: this parent never imports torch or loads libsctl_amd.so, it only waits for the children
    (a plain child process, never an exec) and hands their exit code on.  Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:                      # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this platform
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="laplace_sl", choices=sorted(WORKLOADS) + ["near_apply", "p2p_lists"])
    ap.add_argument("--digits", type=int, default=-1, help="accuracy request; -1 = full precision (the reference default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.workload == "near_apply":
        return bench_near(args)
    if args.workload == "p2p_lists":
        return bench_lists(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # dmabuf IPC is what RCCL needs on this platform; the driver's own `python -m torch.distributed.run ... bench.py` starts the ranks
    # without going through self_launch(), so the rank path sets it too — before torch (and with it the HIP runtime) is loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import sctl_amd
    from sctl_amd.distributed import ShardedDirectSum, slab_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's rank count and --gpus must agree)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: sctl_amd has no CPU path to measure")
    # Rehearsal switch for a ONE-GPU box only: all ranks share device 0 and talk over gloo (RCCL refuses two ranks on one
    # device).  The driver's real multi-GPU runs never set it and use the "nccl" backend (= RCCL over xGMI).
    rehearsal = os.environ.get("SCTL_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible (one rank per GPU)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    kernel, N, dtype, desc = WORKLOADS[args.workload]
    info = sctl_amd.kernel_info(kernel)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    ctx = np.array([7.5, 0.3]) if kernel.startswith("Helmholtz") else None

    # synthetic uniform-random point cloud, identical on every rank (sources are replicated), resident in HBM
    g = torch.Generator(device="cuda").manual_seed(0)
    dev = torch.device("cuda", local_rank)
    r_trg = torch.rand(N * 3, dtype=tdt, device=dev, generator=g)
    r_src = torch.rand(N * 3, dtype=tdt, device=dev, generator=g)
    n_src = torch.rand(N * info["nd"], dtype=tdt, device=dev, generator=g) - 0.5
    v_src = torch.rand(N * info["k0"], dtype=tdt, device=dev, generator=g) - 0.5
    t0, t1 = slab_bounds(N, rank, world)
    out = torch.empty(N * info["k1"], dtype=tdt, device=dev)
    out_slab = torch.empty((t1 - t0) * info["k1"], dtype=tdt, device=dev)
    op = ShardedDirectSum(kernel, ctx=ctx, digits=args.digits)

    # Device time of the evaluation launches, HIP events on the launch stream.  Long steps get an event pair each (the all-gather of a
    # multi-GPU step stays outside it); steps shorter than a few hundred microseconds are bracketed as a whole at N = 1, because two
    # event records per step are themselves a measurable share of such a step.
    kern_ms, gather_ms = [], []
    per_step_events = world > 1 or float(N) * float(N) >= 2.0 ** 34

    def step(timed):
        if timed and per_step_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op.eval_slab(r_trg, r_src, n_src, v_src, out_slab)        # this rank's targets x all sources (HIP kernels)
            e1.record()
            kern_ms.append((e0, e1))
            if world > 1:
                # the collective leg by its own event pair: all-gather of the slabs (the process group makes the current stream wait
                # for it) + the Morton -> caller's order copy; on a rank that finished its slab early it also holds the wait for the others
                e2 = torch.cuda.Event(enable_timing=True)
                res = op.gather(r_trg, out_slab, out)
                e2.record()
                gather_ms.append((e1, e2))
                return res
            return out_slab
        op.eval_slab(r_trg, r_src, n_src, v_src, out_slab)
        return op.gather(r_trg, out_slab, out) if world > 1 else out_slab   # one all-gather, back to the caller's order

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    op.set_targets(r_trg)            # the Morton order of the targets is set-up (cached per target set), like SetTrgCoord: never timed
    if world > 1:                    # RCCL sets its channels up on a collective's first use: that is set-up too, whatever --warmup says
        op.gather(r_trg, out_slab, out)
    for _ in range(args.warmup):
        step(False)
    fence()
    whole = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    tic = time.perf_counter()
    whole[0].record()
    for _ in range(args.steps):
        step(True)
    whole[1].record()
    fence()
    elapsed = time.perf_counter() - tic
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # What every rank measured, and what device it ran on — gathered through the SAME process group the data path uses, so the record
    # shows how many ranks that communicator really had and that they sat on distinct GPUs (outside the timed region).
    per_rank = None
    if world > 1:
        prop = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": prop.name, "uuid": str(getattr(prop, "uuid", "")),
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "slab_targets": t1 - t0,
                "kernel_ms": float(np.mean([a.elapsed_time(b) for a, b in kern_ms])),
                "gather_ms": float(np.mean([a.elapsed_time(b) for a, b in gather_ms]))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    ms_per_step = 1e3 * elapsed / args.steps
    pairs_per_step = float(N) * float(N)                 # all ranks together
    value = pairs_per_step / (elapsed / args.steps)
    fpp = sctl_amd.flops_per_pair(kernel)                # 3 + FLOPS() + 2*SrcDim*TrgDim
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in kern_ms])) if kern_ms else whole[0].elapsed_time(whole[1]) / args.steps
    local_pairs = float(t1 - t0) * float(N)              # pairs ONE launch (this rank) processes
    achieved = local_pairs * fpp / (k_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    plan = sctl_amd.plan(kernel, 0 if dtype == "f64" else 1, t1 - t0, N, args.digits, nt_whole=N)

    if rank == 0:
        line = {
            "metric": "pair-interactions/s, Laplace-SL N x N direct sum" if kernel == "Laplace3D-FxU" else "pair-interactions/s",
            "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "kernel": kernel, "n_trg": N, "n_src": N, "digits": args.digits,
                       "partition": "targets block-partitioned (Morton order) over %d GPU(s), sources replicated, %s" %
                                    (world, "one RCCL all-gather of the potential slabs per step" if world > 1 else "no collective"),
                       "launch": plan},
            # compute-bound: the contract's label for that is "mfma"; on gfx950 the fp64 (fp32) vector pipe this kernel runs on
            # has the same dense peak as the fp64 (fp32) MFMA path, 78.6 (157.3) TFLOP/s, and shares its issue slots
            # (fp32 Laplace SL takes the far pairs' r^2 from the bf16 matrix cores, sctl_amd_eval_pipe; the fraction stays against the fp32 peak)
            "roofline": {"bound": "mfma", "pipe": ("fp64 VALU" if dtype == "f64" else "fp32 VALU") if plan["pipe"] == "vector pipe" else
                                                  "bf16 MFMA (r^2%s as split-bf16 contractions, K = 30) + fp32 VALU (v_rsq_f32, accumulate)" % ("" if kernel == "Laplace3D-FxU" else " and the dot product"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": read_traffic(args.workload) if world == 1 else None,
                         "flops_per_pair": fpp, "kernel_ms": k_ms,
                         "note": "vector-FMA bound (SURVEY.md §8d): algorithmic flops = pairs x (3 + FLOPS() + 2 K0 K1); "
                                 "HBM traffic is ~6e-5 B/pair; achieved is per GPU from HIP-event kernel time"},
            "sctl_gflops": value * info["flops"] / 1e9,   # the reference's own Profile convention (generic-kernel.txx:188)
            "pct_of_peak_all_gpus": 100.0 * value * fpp / (peak * 1e12 * world),
        }
        if world > 1:
            slowest = max(r["kernel_ms"] for r in per_rank)
            line["collective"] = {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(), "rehearsal_shared_gpu": rehearsal,
                "distinct_devices": len({(r["uuid"], r["pci_bus_id"], r["local_rank"]) for r in per_rank}),
                "op": "all_gather_into_tensor of the potential slabs (%d B per rank) + Morton -> caller's order index copy" % ((t1 - t0) * info["k1"] * out.element_size()),
                "gather_ms": max(r["gather_ms"] for r in per_rank),          # slowest rank's collective leg, mean over the timed steps
                "gather_ms_min": min(r["gather_ms"] for r in per_rank),      # the rank that waited least: the collective + copy themselves
                "per_rank": per_rank}
            # the same problem on ONE such GPU would take about this long (work is linear in the slab): compare with the N = 1 record
            line["n1_equiv_ms"] = slowest * world
            line["kernel_ms_slowest_rank"] = slowest
        if world == 1 and args.digits < 0 and dtype == "f64":
            # Informational, outside the timed region and not part of `value`: the same workload at the accuracy the reference's
            # own callers ask for — ParticleFMM defaults to 10 digits (fmm-wrapper.txx:204), BoundaryIntegralOp to tol 1e-10 (:500).
            full = op.eval_slab(r_trg, r_src, n_src, v_src).clone()          # the full-precision result, to measure the 10-digit one against
            op10 = ShardedDirectSum(kernel, ctx=ctx, digits=10)
            op10.eval_slab(r_trg, r_src, n_src, v_src, out_slab)
            torch.cuda.synchronize()
            rel10 = float((out_slab - full).norm() / full.norm())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps10 = max(2, min(200, int(50.0 / ms_per_step)))               # two launches of the headline size, more of a short step
            for _ in range(reps10):
                op10.eval_slab(r_trg, r_src, n_src, v_src, out_slab)
            e1.record()
            torch.cuda.synchronize()
            ms10 = e0.elapsed_time(e1) / reps10
            line["at_reference_callers_accuracy"] = {"digits": 10, "ms_per_step": ms10, "value": pairs_per_step / (ms10 * 1e-3),
                                                     "frac": pairs_per_step * fpp / (ms10 * 1e-3) / 1e12 / peak,
                                                     "rel_l2_vs_full_precision": rel10,
                                                     "per_pair_error_bound": "one Newton step from the v_rsq_f64 seed (relative error d <= 2^-24.18, measured): 1/r comes out low by "
                                                                             "3/2 d^2 <= 4.3e-15 relative; the MEAN of that one-sided error (-1.7e-16) is folded into the per-target scale, "
                                                                             "leaving mean +5e-17, rms 3.1e-16, range [-4.1e-15, +4.7e-16] (profiles/r03_rsq_refine_accuracy.txt), for every "
                                                                             "digits in 8..14; the default (full-precision) mode measures max 2.4 ulp, rms 0.66 ulp per pair, the reference's own default max 2.5 ulp, rms 0.77 ulp"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(kernel, N, dtype)
            except Exception as e:   # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "pair-interactions/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
