#!/bin/bash
# A/B on ONE box: the tile-centred kernel with the XCD-aware (tile, split) mapping (the shipped library) against a build with the
# plain blockIdx mapping (tools/ab/libsctl_amd_plain.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_PLAIN_MAP" OUT=... OBJDIR=...).
for rep in 1 2; do
  for lib in shipped plain; do
    if [ $lib = plain ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_plain.so; else unset SCTL_AMD_LIB; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step']))"
  done
done
