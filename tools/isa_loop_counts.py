#!/usr/bin/env python3
"""Instruction mix of the all-pairs kernels' hot loops, from the device assembly (no GPU needed): for every built-in kernel, fp64, two targets per
lane, full precision (MODE 2) and the 10-digit mode (MODE 1), the speculative (unmasked, constant-trip, unrolled) tile loop of eval_kernel:
fp64 VALU instructions, v_rsq_f64, other VALU and LDS reads PER PAIR, and the issue cycles per wave-pair they cost at the measured rates
(4.1 cycles per fp64 instruction, 16 per v_rsq_f64, 4 per other VALU instruction).
    python tools/isa_loop_counts.py [kernel ...]  > profiles/rNN_isa_loop_counts.json
tools/time_all_digits.py reads the newest such file to print the share of the fp64 issue slots a measured time corresponds to."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["Laplace3D_FxU", "Laplace3D_DxU", "Laplace3D_FxdU", "Stokes3D_FxU", "Stokes3D_DxU", "Stokes3D_FxT", "Stokes3D_FSxU", "Stokes3D_FxUP", "Laplace3D_FDxUdU",
           "Helmholtz3D_FxU"]


def loops(body):
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    out = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def count(body, a, b):
    ins = [l.split()[0] for l in body[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    rsq = sum(v for k, v in c.items() if 'rsq_f64' in k)
    f64 = sum(v for k, v in c.items() if 'f64' in k) - rsq
    other = sum(v for k, v in c.items() if k.startswith('v_') and 'f64' not in k)
    lds = sum(v for k, v in c.items() if k.startswith('ds_read'))
    branches = sum(v for k, v in c.items() if k.startswith('s_cbranch') or k.startswith('s_and_saveexec'))
    return dict(n=len(ins), f64=f64, rsq=rsq, other_valu=other, lds_reads=lds, branches=branches, ldexp=c.get('v_ldexp_f64', 0),
                lds_b64=c.get('ds_read_b64', 0))


def main():
    res = {}
    for k in (sys.argv[1:] or KERNELS):
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                            "--offload-device-only", "-S", os.path.join(ROOT, "sctl_amd", "csrc", "inst_%s.hip" % k), "-o", asm], check=True, stderr=subprocess.DEVNULL)
            src = open(asm).read()
        for mode in (2, 1):
            sym = "_ZN8sctl_amd11eval_kernelINS_%d%sEdLi%dELi2EEEvNS_8EvalArgsIT0_EE" % (len(k), k, mode)
            i0 = src.index("\n" + sym + ":")
            body = src[i0:src.index(".Lfunc_end", i0)].split("\n")
            # candidates: innermost single-branch loops with 4 v_rsq_f64 (two sources x two targets per trip); the speculative pass has no
            # masking select (v_cndmask) in it; a kernel with launch-uniform variants has several: the first one WITH the feature the
            # default context uses (Helmholtz: complex wavenumber, one-reduction form = a period-factor read, no v_ldexp)
            cands = []
            for a, b in loops(body):
                c = count(body, a, b)
                if c["rsq"] >= 2 and c["branches"] == 1 and not any("v_cndmask" in l for l in body[a:b + 1]):
                    cands.append(c)
            if k == "Helmholtz3D_FxU":
                cands = [c for c in cands if c["ldexp"] == 0 and c["lds_b64"] >= c["rsq"]] or cands
            most = max(c["rsq"] for c in cands)               # the fully unrolled body (one v_rsq_f64 per pair)
            c = next(c for c in cands if c["rsq"] == most)
            per = {kk: c[kk] / float(most) for kk in ("f64", "rsq", "other_valu", "lds_reads")}
            per["issue_cycles_per_wave_pair"] = 4.1 * per["f64"] + 16.0 * per["rsq"] + 4.0 * per["other_valu"]
            res["%s/mode%d" % (k.replace("_", "-", 1), mode)] = per
    json.dump({"what": "per PAIR, speculative tile loop of eval_kernel<K, double, MODE, T=2>, gfx950, from hipcc -S", "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
