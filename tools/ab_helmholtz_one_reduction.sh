#!/bin/bash
# A/B on ONE box: Helmholtz all-pairs kernel with ONE reduction of r for e^{ikr} (the shipped library) against the round-2 form with two
# (tools/ab/libsctl_amd_helm2red.so: make -C sctl_amd/csrc EXTRA=-DSCTL_AMD_EXP_HELMHOLTZ_TWO_REDUCTIONS OUT=$PWD/tools/ab/libsctl_amd_helm2red.so OBJDIR=/tmp/helm2red_build).
for rep in 1 2; do
  for lib in one_reduction two_reductions; do
    if [ $lib = two_reductions ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_helm2red.so; else unset SCTL_AMD_LIB; fi
    echo "== $lib rep $rep"
    python3 tools/time_helmholtz.py 2>&1 | grep float64
  done
done
