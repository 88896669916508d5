#!/bin/bash
# Profiles `python bench.py` on the GPU box: per-kernel time (kernel-trace + stats) and, in SEPARATE passes,
# the HBM read / write counters.  Run from the repo root through gpurun; outputs land in gpurun_out/prof_<tag>/.
#   tools/profile_bench.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$OLDPWD/bench.py" $args > "$out/bench_under_trace.json" 2> "$out/trace.log" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$OLDPWD/bench.py" $args > /dev/null 2> "$out/pmc_fetch.log" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$OLDPWD/bench.py" $args > /dev/null 2> "$out/pmc_write.log" &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- python3 "$OLDPWD/bench.py" $args > /dev/null 2> "$out/pmc_sq.log" &&
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_clk" -- python3 "$OLDPWD/bench.py" $args > /dev/null 2> "$out/pmc_clk.log"
rc=$?
cd "$OLDPWD"
find "$out" -name "*.csv" | wc -l
exit $rc
