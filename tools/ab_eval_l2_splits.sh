#!/bin/bash
# A/B: the exact all-pairs kernel with source splits of <= 2 MB in multiples of 8, each owned by one XCD (the shipped
# library) against the round-2 plan — few, large splits; the XCD-aware mapping only where the splits already come in eights —
# (tools/ab/libsctl_amd_nol2splits.so: make -C sctl_amd/csrc EXTRA="-DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_NO_L2_SPLITS" OUT=... OBJDIR=...).  Per workload and library: step time from
# bench.py, and FETCH_SIZE / WRITE_SIZE of the dominant kernel from two rocprofv3 --pmc passes.
out=$PWD/gpurun_out/ab_l2splits; mkdir -p $out
root=$PWD
for w in stokeslet laplace_sldl helmholtz; do
  for lib in shipped nol2splits; do
    if [ $lib = nol2splits ]; then export SCTL_AMD_LIB=$root/tools/ab/libsctl_amd_nol2splits.so; else unset SCTL_AMD_LIB; fi
    python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib: %.2f ms/step  kernel %.2f ms  frac %.4f  plan %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['launch']))"
    for c in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $c --output-format csv -d $out/${w}_${lib}_$c -- python3 $root/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/${w}_${lib}_$c.log)
      python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$out/${w}_${lib}_$c/*/*_counter_collection.csv"), key=lambda p: __import__("os").path.getmtime(p))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "eval_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("   $w $lib %s: %.1f MB per launch (raw KB mean %.1f over %d launches)%s" % (k, sum(v) / len(v) * 1024 / 1e6 * (2 if k == "FETCH_SIZE" else 1), sum(v) / len(v), len(v), "  [x2 gfx950 correction applied]" if k == "FETCH_SIZE" else ""))
PY
    done
  done
done
