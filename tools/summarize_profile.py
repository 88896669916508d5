#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into the files committed under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / average ns)
  profiles/<tag>_pmc_summary.csv    per-kernel means of the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*; one pass each)
  profiles/hbm_traffic.json         HBM bytes per launch of the dominant kernel, read by bench.py for roofline.traffic
Also derives, for the dominant kernel: the shader clock it ran at (GRBM_GUI_ACTIVE / 8 XCDs / kernel time, MI355X_MICROARCH.md "DVFS give-back")
and the VALU instructions per wave.
usage: tools/summarize_profile.py <tag> <workload> [<substring the dominant kernel's name must contain>]"""
import collections
import csv
import glob
import json
import os
import sys

tag, workload = sys.argv[1], sys.argv[2]
P = "gpurun_out/prof_" + tag
q = lambda name: '"%s"' % name if "," in name else name      # kernel names with template arguments carry commas
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for pre in ("void ", "sctl_amd::", "rocprim::ROCPRIM_400200_NS::detail::"):
        name = name.replace(pre, "")
    return name[:90]


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    found = glob.glob(pattern)
    return max(found, key=os.path.getmtime) if found else None


rows = list(csv.DictReader(open(newest(P + "/trace/runc/*_kernel_stats.csv"))))
with open("profiles/%s_kernel_stats.csv" % tag, "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats of: python bench.py --steps 3 --warmup 1 --no-cpu-baseline (workload %s)\n" % workload)
    fh.write("kernel,calls,total_ns,average_ns,percentage\n")
    for r in rows:
        fh.write("%s,%s,%s,%s,%s\n" % (q(short(r["Name"])), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
want = sys.argv[3] if len(sys.argv) > 3 else ""
dominant = max((r for r in rows if want in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
summ = collections.defaultdict(dict)
meta = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_clk"):
    agg = collections.defaultdict(list)
    found = newest("%s/%s/runc/*_counter_collection.csv" % (P, sub))
    if not found:
        continue
    for r in csv.DictReader(open(found)):
        if "at::native" in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        meta[k] = "grid=%s wg=%s lds=%s vgpr=%s sgpr=%s" % (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    for (k, c), v in agg.items():
        summ[k][c] = (len(v), sum(v) / len(v))
with open("profiles/%s_pmc_summary.csv" % tag, "w") as fh:
    fh.write("# rocprofv3 --pmc passes, FETCH_SIZE / WRITE_SIZE / SQ set each in its OWN run without tracing; mean per launch\n")
    fh.write("kernel,launch_geometry,counter,launches,mean_per_launch\n")
    for k, d in summ.items():
        for c, (n, m) in sorted(d.items()):
            fh.write("%s,%s,%s,%d,%.6g\n" % (q(k), meta[k], c, n, m))
dk = short(dominant["Name"])
fetch_kb, write_kb = summ[dk]["FETCH_SIZE"][1], summ[dk]["WRITE_SIZE"][1]
path = "profiles/hbm_traffic.json"
data = json.load(open(path)) if os.path.exists(path) else {}
import re
clk = summ[dk].get("GRBM_GUI_ACTIVE")
waves, valu = summ[dk].get("SQ_WAVES"), summ[dk].get("SQ_INSTS_VALU")
data[workload] = {
    "kernel": dk, "kernel_average_ms": float(dominant["AverageNs"]) / 1e6,
    "symbol_tokens": [t for t in re.findall(r"[A-Za-z_][A-Za-z_0-9]{5,}", dk) if t not in ("double", "float")],   # bench.py checks them against the built library
    "shader_clock_GHz": (clk[1] / 8 / float(dominant["AverageNs"])) if clk else None,
    "valu_insts_per_wave": (valu[1] / waves[1]) if (waves and valu) else None,
    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "hbm_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> read side doubled (an upper bound "
                  "here: the tile-fill loads are 8/16 B per lane at a 24-B stride, not the calibrated 16 B/lane stream); WRITE_SIZE exact",
    "source": "profiles/%s_pmc_summary.csv" % tag}
json.dump(data, open(path, "w"), indent=1)
print(dk, "avg ms", data[workload]["kernel_average_ms"], "traffic", data[workload]["hbm_bytes_per_launch"])
