#!/bin/bash
# A/B on ONE box: the headline workload with the shipped targets per lane of the fp64 tile-centred kernels (four = 256 per wave since late round 3; two when
# profiles/r03_ab_centered_T.txt was measured) against another count — tools/ab/libsctl_amd_T4.so: make -C sctl_amd/csrc
# EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_CENTERED_T=4" OUT=... OBJDIR=... (likewise 2, 3, 6, 8; edit the list below)
for rep in 1 2; do
  for lib in shipped T4; do
    if [ $lib = shipped ]; then unset SCTL_AMD_LIB; else export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_$lib.so; fi
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms  rel-L2 of the 10-digit result %.1e' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step'], d['at_reference_callers_accuracy']['rel_l2_vs_full_precision']))"
  done
done
