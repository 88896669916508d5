#!/bin/bash
# A/B on ONE box: the tile-centred kernel as shipped (4 far records per unrolled group, the 4 waves per SIMD its registers allow) against builds that ask
# the compiler for 5 or 6 waves per SIMD and / or unroll by 2 or 8 (tools/ab/libsctl_amd_{W5,W6,U2,U8,W5U2}.so:
# make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_CENTERED_WAVES=5" OUT=... OBJDIR=...; likewise -DSCTL_AMD_EXP_FAR_UNR=2).
for rep in 1 2; do
  for lib in shipped W5 W6 U2 U8 W5U2; do
    if [ $lib = shipped ]; then unset SCTL_AMD_LIB; else export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_$lib.so; fi
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: %.2f ms  frac %.4f  10-digit %.2f ms' % (d['ms_per_step'], d['roofline']['frac'], d['at_reference_callers_accuracy']['ms_per_step']))"
  done
done
