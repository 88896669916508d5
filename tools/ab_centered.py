"""A/B timing inside ONE gpurun call (same GPU): exact kernel vs the tile-centred path, interleaved.
usage (on the GPU box): python tools/ab_centered.py [log2 N ...]"""
# NOTE: the SCTL_AMD_EXPERIMENT_* switches exist only in a library built with `make -C sctl_amd/csrc EXTRA=-DSCTL_AMD_EXPERIMENTS OUT=... OBJDIR=...`
# (point SCTL_AMD_LIB at it); the shipped libsctl_amd.so ignores them.
import os
import subprocess
import sys

cfgs = [("exact", {"SCTL_AMD_CENTERED": "0"}), ("centred", {}), ("centred, 1 split", {"SCTL_AMD_EXPERIMENT_SPLITS": "1"}),
        ("centred, 4 splits", {"SCTL_AMD_EXPERIMENT_SPLITS": "4"}), ("centred, all far (wrong)", {"SCTL_AMD_EXPERIMENT_NEAR_FACTOR": "0"})]
for logn in sys.argv[1:] or ["20"]:
    for rep in range(2):
        for name, env in cfgs:
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, "tools/time_one.py", "Laplace3D-FxU", logn, "f64"], env=e, capture_output=True, text=True).stdout.strip()
            print("%-26s %s" % (name, out[60:]), flush=True)
