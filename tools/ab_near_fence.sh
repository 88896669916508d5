#!/bin/bash
# A/B on ONE box: the headline workload with the near pairs fenced one after the other (shipped; centered_kernel.hpp: flush_near) against a build without the
# fence (tools/ab/libsctl_amd_nofence.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_NO_NEAR_FENCE" OUT=... OBJDIR=...).
for rep in 1 2; do
  for lib in shipped nofence; do
    if [ $lib = shipped ]; then unset SCTL_AMD_LIB; else export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_$lib.so; fi
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step']))"
  done
done
