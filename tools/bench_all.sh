set -u
mkdir -p gpurun_out/${TAG:-r04}_bench
python bench.py > gpurun_out/${TAG:-r04}_bench/n1.json 2> gpurun_out/${TAG:-r04}_bench/n1.err
python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG:-r04}_bench/n1_driver_style.json 2>> gpurun_out/${TAG:-r04}_bench/n1.err
for w in laplace_sl_16k laplace_sldl stokeslet stokeslet_f32 helmholtz p2p_lists near_apply; do
  st=5; [ $w = laplace_sl_16k ] && st=200
  python bench.py --workload $w --steps $st --warmup 2 > gpurun_out/${TAG:-r04}_bench/$w.json 2>> gpurun_out/${TAG:-r04}_bench/n1.err
done
python bench.py --workload laplace_sl_f32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG:-r04}_bench/laplace_sl_f32_1gpu.json 2>> gpurun_out/${TAG:-r04}_bench/n1.err
python tools/time_all_digits.py > gpurun_out/${TAG:-r04}_bench/all_kernels_full_vs_10digits.txt 2>&1
python tools/rank_share.py > gpurun_out/${TAG:-r04}_bench/rank_share.txt 2>&1
for f in gpurun_out/${TAG:-r04}_bench/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "%.3f ms/step" % d['ms_per_step'], "frac %.4f" % d['roofline']['frac'], "value %.4g" % d['value'], (d.get('at_reference_callers_accuracy') or {}).get('ms_per_step'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'), (d.get('cpu_baseline') or {}).get('threads'))
PY
done
cat gpurun_out/${TAG:-r04}_bench/all_kernels_full_vs_10digits.txt gpurun_out/${TAG:-r04}_bench/rank_share.txt
