#!/usr/bin/env python3
"""The objects that REFUTED the first suspect for the run-to-run different near sums of round 3 (profiles/r04_near_fault_report.md): code objects
of the no-fence build of centered_mfma_f32_kernel<true, 4> (profiles/r04_near_fault/kernel_nofence.s) in which ONLY the places where a vector instruction overwrites the source register of the transcendental instruction directly in front of it are changed
(centered_mfma_f32_kernel<true, 4>: `v_rsq_f32 v34, v36 ; v_pk_mul_f32 v[36:37], ...` in the near loop, `v_rsq_f32 v12, v4 ; v_mov_b32 v4, v27` in its odd tail,
and the same two in the final flush), found by tools/check_isa_rules.py's listing:
    war_w<N>      N wait states (s_nop) between the transcendental and the overwriting instruction, N = 1, 2, 4, 8, 16, 32, at all sites
    war_w16_mid   16 wait states at the two sites of the flush BETWEEN tiles only
    war_ren       the overwriting instruction's destination (and its readers) renamed to registers nobody else uses: the source is never overwritten, no idle instruction
    war_ren_mid   the same at the two between-tiles sites only
    war_burst     nothing changed at the sites; `s_nop 1` behind each v_rsq_f32 of the far loop's 16-instruction bursts (thins the other waves' load on the unit)
Result (profiles/r04_kernel_repeat_war.txt): every one of them is as faulty as the unpatched kernel, the renamed ones included: the overwritten source is NOT the cause.
part 1 (no GPU):  python tools/kernel_repeat_war.py build       part 2 (GPU box):  python tools/kernel_repeat_war.py run   (24 launches per object, tools/ubench/kernel_repeat)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_rules as cth

D = os.path.join(ROOT, "tools", "ab", "repeat")
LLVM = "/opt/rocm/lib/llvm/bin"
SYM = "_ZN8sctl_amd24centered_mfma_f32_kernelILb1ELi4EEEvNS_8EvalArgsIfEE"
VARIANTS = ["war_w1", "war_w2", "war_w4", "war_w8", "war_w16", "war_w32", "war_w16_mid", "war_ren", "war_ren_mid", "war_burst"]


def assemble(name, text):
    s, o, co = (os.path.join(D, name + e) for e in (".s", ".o", ".co"))
    open(s, "w").write(text)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", o, "-o", co], check=True)
    os.remove(o)
    os.remove(s)


def nops(n):
    out = []
    while n > 0:
        k = min(n, 8)
        out.append("\ts_nop %d" % (k - 1))
        n -= k
    return out


def rename(body, w, new_base):
    """the instruction at line w writes registers D; give them the names new_base.. and rename their readers up to the next write of each"""
    op, wr, rd, _ = cth.parse(body[w])
    old = sorted(wr)
    m = {r: new_base + k for k, r in enumerate(old)}

    def sub_operand(tok, regs):
        def rng(mm):
            a, b = int(mm.group(1)), int(mm.group(2))
            if a in regs and b in regs:
                return "v[%d:%d]" % (regs[a], regs[b])
            if a in regs and b == a + 1:   # a broadcast read (op_sel_hi 0) of the renamed low register through a register pair: the high register's value is not used
                return "v[%d:%d]" % (regs[a], regs[a] + 1)
            assert a not in regs and b not in regs, "partial overlap: " + tok
            return mm.group(0)
        tok = re.sub(r"\bv\[(\d+):(\d+)\]", rng, tok)
        return re.sub(r"\bv(\d+)\b", lambda mm: "v%d" % regs[int(mm.group(1))] if int(mm.group(1)) in regs else mm.group(0), tok)

    def split(line):
        t = line.strip()
        head, rest = t.split(None, 1)
        return head, [x.strip() for x in rest.split(",")]
    head, ops = split(body[w])
    ops[0] = sub_operand(ops[0], m)
    body[w] = "\t" + head + " " + ", ".join(ops)
    live = dict(m)
    i = w + 1
    changed = [w]
    def dead_from(j, r, why):
        for k in range(j, min(j + 400, len(body))):
            q = cth.parse(body[k])
            if q is None:
                continue
            assert r not in q[2], "v%d is still read at line %d (%s)" % (r, k, why)
            if r in q[1] or q[0].startswith(("s_branch", "s_endpgm")):
                return

    while live and i < len(body):
        p = cth.parse(body[i])
        if p is None:
            i += 1
            continue
        op2, wr2, rd2, _ = p
        if op2.startswith(("s_cbranch", "s_branch")):   # the renamed values must be dead here: along the branch and behind it the first access of each register is a write
            tgt = body[i].split()[-1]
            j = next(k for k, l in enumerate(body) if l.startswith(tgt + ":"))
            for r in live:
                dead_from(j, r, "behind the branch of line %d" % i)
                if not op2.startswith("s_branch"):
                    dead_from(i + 1, r, "falling through line %d" % i)
            break
        if rd2 & set(live):
            head, ops = split(body[i])
            start = 0 if op2.startswith(("v_cmp", "ds_write", "global_store", "buffer_store")) else 1
            for k in range(start, len(ops)):
                ops[k] = sub_operand(ops[k], live)
            body[i] = "\t" + head + " " + ", ".join(ops)
            changed.append(i)
        for r in list(live):
            if r in wr2:
                del live[r]
        i += 1
    return changed


def build():
    os.makedirs(D, exist_ok=True)
    src = open(os.path.join(ROOT, "profiles", "r04_near_fault", "kernel_nofence.s")).read()   # the frozen reproducer: that kernel as round 4's compiler made it
    i0 = src.index("\n" + SYM + ":") + 1
    i1 = src.index(".Lfunc_end", i0)
    body0 = src[i0:i1].split("\n")
    _, rule1, _ = cth.check(body0, 4)
    sites = [(o, i) for o, ol, i, il, d in rule1 if d <= 2 and il.startswith("v_") and not il.startswith(cth.TRANS + ("v_mfma",))]
    print("sites (transcendental line, overwriting line) in %s:" % SYM)
    for o, i in sites:
        print("   %5d  %-34s %5d  %s" % (o, body0[o].strip(), i, body0[i].strip()))
    assert len(sites) == 4, "expected the four sites of the round-3 build"
    mfma_lines = [i for i, l in enumerate(body0) if "v_mfma" in l]
    mid = [s for s in sites if s[0] < mfma_lines[0]]          # the flush between tiles sits in front of the far loop
    assert len(mid) == 2

    def emit(name, body, extra_vgprs=0):
        text = src[:i0] + "\n".join(body) + src[i1:]
        if extra_vgprs:   # the renamed registers lie between the kernel's last vector register and its (unused) accumulation-register offset
            k0 = text.index(".amdhsa_kernel " + SYM)
            k1 = text.index(".end_amdhsa_kernel", k0)
            desc = text[k0:k1]
            nf = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
            acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1))
            assert nf + extra_vgprs <= acc, (nf, acc)
            desc = desc.replace(".amdhsa_next_free_vgpr %d" % nf, ".amdhsa_next_free_vgpr %d" % (nf + extra_vgprs))
            text = text[:k0] + desc + text[k1:]
            text = re.sub(r"(\.vgpr_count:\s+)%d(\n(?:.*\n){0,12}?.*\.name:\s+%s)" % (nf, SYM), lambda mm: mm.group(1) + str(nf + extra_vgprs) + mm.group(2), text)
        assemble(name, text)

    def with_nops(sel, n):
        ins = {o: n for o, i in sel}
        out = []
        for k, l in enumerate(body0):
            out.append(l)
            if k in ins:
                out.extend(nops(ins[k]))
        return out
    for n in (1, 2, 4, 8, 16, 32):
        emit("war_w%d" % n, with_nops(sites, n))
    emit("war_w16_mid", with_nops(mid, 16))
    nf = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", src[src.index(".amdhsa_kernel " + SYM):]).group(1))
    for name, sel in (("war_ren", sites), ("war_ren_mid", mid)):
        b = list(body0)
        for o, i in sel:
            ch = rename(b, i, nf)
            print("   %s: line %d and %d reader line(s) renamed to v%d.." % (name, i, len(ch) - 1, nf))
        emit(name, b, 2)
    # the far loop's bursts: runs of >= 8 consecutive v_rsq_f32
    out, run = [], 0
    burst = set()
    for k, l in enumerate(body0):
        if l.strip().startswith("v_rsq_f32"):
            run += 1
        else:
            if run >= 8:
                burst.update(range(k - run, k))
            if cth.parse(l) is not None:
                run = 0
    for k, l in enumerate(body0):
        out.append(l)
        if k in burst:
            out.append("\ts_nop 1")
    print("   war_burst: %d v_rsq_f32 of the far loop's bursts followed by s_nop 1" % len(burst))
    emit("war_burst", out)
    emit("nofence", body0)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "ubench", "kernel_repeat.cpp"), "-o",
                    os.path.join(ROOT, "tools", "ubench", "kernel_repeat")], check=True)
    print(sorted(os.listdir(D)))


def run():
    exe = os.path.join(ROOT, "tools", "ubench", "kernel_repeat")
    print("# tools/kernel_repeat_war.py run: 24 launches of centered_mfma_f32_kernel<true, 4> per code object on one problem (2^17 targets x 2^16 sources, fp32), another kernel", flush=True)
    print("# between any two launches (POISON=1); a run counts as off when any of its values differs from the value most runs give", flush=True)
    env = dict(os.environ, POISON="1")
    for v in ["nofence"] + VARIANTS + ["nofence"]:
        subprocess.run([exe, os.path.relpath(os.path.join(D, v + ".co"), ROOT)], env=env, cwd=ROOT, check=True)
        sys.stdout.flush()


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
