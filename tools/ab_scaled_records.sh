#!/bin/bash
# A/B on ONE box: the kernels with several powers of 1/r (Stokeslet family, fused Laplace) on the unnormalised 1/r with the density kept twice in
# the record (the shipped library) against the build before that change (tools/ab/libsctl_amd_prev.so: the previous commit's sctl_amd/csrc built
# with OUT=...).  Helmholtz and the one-power kernels are the same code in both: they show the run-to-run noise.
for rep in 1 2; do
  for lib in shipped prev; do
    if [ $lib = prev ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_prev.so; else unset SCTL_AMD_LIB; fi
    echo "== $lib rep $rep: all kernels at 2^18, full precision and 10 digits"
    python3 tools/time_all_digits.py 2>/dev/null
  done
done
for lib in shipped prev; do
  if [ $lib = prev ]; then export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_prev.so; else unset SCTL_AMD_LIB; fi
  for w in stokeslet laplace_sldl; do python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w: %.2f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; done
done
