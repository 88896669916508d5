"""Build libsctl_amd.so in-tree with hipcc for gfx950 (sctl_amd/csrc/Makefile)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_library(jobs=None, verbose=False):
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    path = os.path.join(_HERE, "libsctl_amd.so")
    if not os.path.exists(path):
        raise RuntimeError("build did not produce " + path)
    return path
