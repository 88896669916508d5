"""Time GenericKernel::KernelMatrix on the device: python tools/time_matrix.py"""
import sys

import torch

sys.path.insert(0, '.')
import sctl_amd  # noqa: E402

for name, Nt, Ns in (("Laplace3D-FxU", 16384, 16384), ("Stokes3D-FxU", 8192, 4096), ("Stokes3D-DxU", 8192, 4096), ("Laplace3D-FxdU", 16384, 8192)):
    info = sctl_amd.kernel_info(name)
    g = torch.Generator(device='cuda').manual_seed(0)
    xt = torch.rand(Nt * 3, dtype=torch.float64, device='cuda', generator=g)
    xs = torch.rand(Ns * 3, dtype=torch.float64, device='cuda', generator=g)
    xn = torch.rand(Ns * info['nd'], dtype=torch.float64, device='cuda', generator=g) - 0.5
    M = sctl_amd.kernel_matrix_device(name, xt, xs, xn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        sctl_amd.kernel_matrix_device(name, xt, xs, xn, M=M)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = M.numel() * 8 / 1e9
    print(f"KernelMatrix {name:16s} Nt={Nt} Ns={Ns}: {ms:8.3f} ms  {gb:6.2f} GB written  {gb / ms * 1e3:7.1f} GB/s  {Nt * Ns / ms / 1e6:8.1f} Gpairs/s", flush=True)
