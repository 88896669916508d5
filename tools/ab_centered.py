"""A/B timing inside ONE gpurun call (same GPU): exact kernel vs the tile-centred path, interleaved."""
import os, subprocess, sys
cfgs = [("exact", {"SCTL_AMD_CENTERED": "0"}), ("centred S1", {}), ("centred S2", {"SCTL_AMD_EXPERIMENT_SPLITS": "2"}),
        ("centred S4", {"SCTL_AMD_EXPERIMENT_SPLITS": "4"}), ("centred S8", {"SCTL_AMD_EXPERIMENT_SPLITS": "8"}),
        ("centred S16", {"SCTL_AMD_EXPERIMENT_SPLITS": "16"}), ("centred S4 all-far", {"SCTL_AMD_EXPERIMENT_SPLITS": "4", "SCTL_AMD_EXPERIMENT_NEAR_FACTOR": "0"})]
for logn in sys.argv[1:] or ["20"]:
    for rep in range(2):
        for name, env in cfgs:
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, "tools/time_one.py", "Laplace3D-FxU", logn, "f64"], env=e, capture_output=True, text=True).stdout.strip()
            print("%-20s %s" % (name, out[60:]), flush=True)
