"""Launch-bound sizes: the device-resident entry captured in a HIP graph (torch.cuda.CUDAGraph) against plain stream launches.
The entry allocates nothing after its first call on a stream (workspace.hpp), so it can be captured as it is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sctl_amd

def bench(name, N, reps=200):
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device="cuda").manual_seed(0)
    xt = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g); xs = torch.rand(N * 3, dtype=torch.float64, device="cuda", generator=g)
    xn = torch.rand(N * info["nd"], dtype=torch.float64, device="cuda", generator=g) - 0.5
    f = torch.rand(N * info["k0"], dtype=torch.float64, device="cuda", generator=g) - 0.5
    v = torch.zeros(N * info["k1"], dtype=torch.float64, device="cuda")
    def step():
        v.zero_()
        sctl_amd.eval_device(name, xt, xs, xn, f, v_trg=v)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()                # warm-up on the capture stream: scratch block allocated here
        s.synchronize()
        ref = v.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            step()
        graph.replay(); s.synchronize()
        same = torch.equal(v, ref)
        t0 = time.perf_counter()
        for _ in range(reps): step()
        s.synchronize()
        t_plain = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(reps): graph.replay()
        s.synchronize()
        t_graph = (time.perf_counter() - t0) / reps * 1e6
    print("%-16s N=%6d  stream launches %7.1f us/step   graph replay %7.1f us/step   identical result: %s" % (name, N, t_plain, t_graph, same), flush=True)

for N in (1 << 10, 1 << 12, 1 << 14, 1 << 16):
    bench("Laplace3D-FxU", N)
bench("Stokes3D-DxU", 1 << 12)
