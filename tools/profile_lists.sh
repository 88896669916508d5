out=$PWD/gpurun_out/prof_lists; mkdir -p $out; root=$PWD
export LISTS_KERNELS=Laplace3D-FxU
for g in 32 64; do
  export LISTS_GRIDS=$g
  python3 tools/time_lists.py 2>&1 | grep -v amdgpu
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$g -- python3 $root/tools/time_lists.py > /dev/null 2> $out/trace_$g.log)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_$g -- python3 $root/tools/time_lists.py > /dev/null 2> $out/pmc_$g.log)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2_$g -- python3 $root/tools/time_lists.py > /dev/null 2> $out/pmc2_$g.log)
  python3 - <<PY
import csv, glob, collections, os
for sub in ("trace_$g", "pmc_$g", "pmc2_$g"):
    fs = glob.glob("$out/%s/*/*_counter_collection.csv" % sub) + glob.glob("$out/%s/*/*_kernel_stats.csv" % sub)
    for f in fs:
        rows = list(csv.DictReader(open(f)))
        if "Counter_Name" in rows[0]:
            agg = collections.defaultdict(list)
            for r in rows:
                if "lists_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items(): print("grid $g", k, "%.4g" % (sum(v) / len(v)), "n", len(v))
        else:
            for r in rows:
                if "lists_kernel" in r["Name"]: print("grid $g", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
