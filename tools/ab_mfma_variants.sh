#!/bin/bash
# A/B on ONE box: variants of the fp32 matrix-core kernels (centered_mfma_kernel.hpp): the double layer's VALU part as a plain loop (DLB0), with the cube
# per pair of values (DLB1) or in half-batches (shipped) — tools/ab/libsctl_amd_DLB{0,1}.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS
# -DSCTL_AMD_EXP_DL_BATCH=0" OUT=... OBJDIR=... —, and 256 instead of 128 targets per wave (SCTL_AMD_MFMA_CB=8).
for lib in shipped DLB0 DLB1; do
  if [ $lib = shipped ]; then unset SCTL_AMD_LIB; else export SCTL_AMD_LIB=$PWD/tools/ab/libsctl_amd_$lib.so; fi
  echo "== $lib"; python3 tools/ab_mfma_f32.py --dl --mfma-only 20 21 2>/dev/null
done
unset SCTL_AMD_LIB
for cb in 4 8; do
  export SCTL_AMD_MFMA_CB=$cb
  echo "== shipped, $((cb * 32)) targets per wave"; python3 tools/ab_mfma_f32.py --mfma-only 20 21 2>/dev/null; python3 tools/ab_mfma_f32.py --dl --mfma-only 20 21 2>/dev/null
done
