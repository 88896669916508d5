#!/bin/bash
# A/B on ONE box: full precision (MODE 2) with the four-instruction cubic step (the shipped library) against a build that keeps the
# five-instruction Halley step (tools/ab/libsctl_amd_halley.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_HALLEY" OUT=... OBJDIR=...).
for rep in 1 2; do
  for lib in shipped halley; do
    if [ $lib = halley ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_halley.so; else unset SCTL_AMD_LIB; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step']))"
  done
done
for lib in shipped halley; do
  if [ $lib = halley ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_halley.so; else unset SCTL_AMD_LIB; fi
  echo "== $lib: all kernels at 2^18, full precision and 10 digits"
  python3 tools/time_all_digits.py 2>/dev/null
done
