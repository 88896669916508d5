# PMC counters of the evaluation kernel of ONE timing run (tools/time_one.py arguments), e.g.
#   tools/pmc_one.sh stokes_centred Stokes3D-FxU 18 f64         (environment: SCTL_AMD_CENTERED, SCTL_AMD_LIB as for time_one.py)
# writes gpurun_out/pmc_<tag>.txt: per counter the mean over the kernel's launches (the kernel = the one with the largest total SQ_BUSY_CYCLES)
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag; mkdir -p $out; root=$PWD
python3 tools/time_one.py "$@" 2>&1 | grep -v amdgpu > $out.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --output-format csv -d $out/$n -- python3 $root/tools/time_one.py "$@" > /dev/null 2> $out/$n.log)
done
python3 - $out >> $out.txt <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
k = max(agg, key=lambda n: sum(agg[n].get("SQ_BUSY_CYCLES", [0])) + sum(agg[n].get("GRBM_GUI_ACTIVE", [0])))
print("kernel:", k[:140])
for c, v in sorted(agg[k].items()): print("  %-22s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $out.txt
