#!/bin/bash
# A/B on ONE box: the headline workload with two targets per lane (shipped: 128 per wave) against four (256 per wave: half the per-tile staging per pair, a larger
# cluster) — tools/ab/libsctl_amd_T4.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_CENTERED_T=4" OUT=... OBJDIR=...
for rep in 1 2; do
  for lib in shipped T4; do
    if [ $lib = shipped ]; then unset SCTL_AMD_LIB; else export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_$lib.so; fi
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms  rel-L2 of the 10-digit result %.1e' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step'], d['at_reference_callers_accuracy']['rel_l2_vs_full_precision']))"
  done
done
