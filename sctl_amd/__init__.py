"""sctl_amd — MI355X-native direct kernel summation behind SCTL's kernel-functor interface.

The product is libsctl_amd.so (hand-written HIP for gfx950, C ABI in include/sctl_amd.h) plus the header-only C++
host surface in include/sctl_amd/.  This Python package is plumbing for tests, benchmarks and multi-GPU runs:
ctypes access to the C ABI, torch device memory / streams, and the torch.distributed (RCCL) slab driver.
There is no CPU fallback anywhere in this package: if the library is missing, importing the API raises.
"""
from .api import (DirectOp, GenericKernel, ListsPlan, NearOp, KERNEL_NAMES, eval_lists_host, load_plugin, counters, device_count, init, finalize, eval_device, eval_host, flops_per_pair, kernel_id,  # noqa: F401
                  kernel_info, kernel_matrix_batch_host, kernel_matrix_device, kernel_matrix_host, last_error, lib, library_path, plan, reset_counters)
from .build import build_library  # noqa: F401

__all__ = ["DirectOp", "GenericKernel", "ListsPlan", "NearOp", "KERNEL_NAMES", "eval_lists_host", "load_plugin", "build_library", "counters", "device_count", "init", "finalize", "eval_device", "eval_host", "flops_per_pair",
           "kernel_id", "kernel_info", "kernel_matrix_batch_host", "kernel_matrix_device", "kernel_matrix_host", "last_error", "lib", "library_path", "plan",
           "reset_counters"]
