#!/bin/bash
# One rocprofv3 campaign over the bench workloads (kernel trace + four counter passes each, tools/profile_bench.sh) and the
# summaries the repository keeps under profiles/ (tools/summarize_profile.py).  Run on the GPU box from the repo root:
#   tools/profile_all.sh <tag> [workload ...]
set -u
tag=${1:-r02}; shift || true
wl=${*:-laplace_sl laplace_sl_16k laplace_sldl stokeslet stokeslet_f32 helmholtz p2p_lists near_apply}
for w in $wl; do
  echo "== $w"
  tools/profile_bench.sh ${tag}_$w --workload $w || { echo "profiling $w failed"; exit 1; }
  filter=""; [ "$w" = near_apply ] && filter=near_gemv_kernel     # that bench line also times a far field: the workload's kernel is the GEMV
  python3 tools/summarize_profile.py ${tag}_$w $w $filter || exit 1
  cp gpurun_out/prof_${tag}_$w/bench_under_trace.json profiles/${tag}_${w}_bench_under_trace.json
done
