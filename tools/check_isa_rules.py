#!/usr/bin/env python3
"""Static rules over the device assembly hipcc makes of the shipped kernels (no GPU needed).  What they come from: DESIGN.md §4.2a, profiles/r04_near_fault_report.md.

RULE A (fails the check) — no packed-fp32 instruction in a masked-pair loop of a kernel that issues transcendental bursts.
    The run-to-run different near sums of round 3 (centered_mfma_f32_kernel<true, 4> without the near fence: lanes 48-63 of one target column block, whole contributions
    lost) were pinned down in round 4 with patched copies of that kernel's assembly (tools/kernel_repeat_ranges.py, profiles/r04_kernel_repeat_bisect*.txt):
      * idle instructions cure it if and only if they follow EVERY packed-fp32 (v_pk_*) AND every transcendental instruction of the near flush and every transcendental
        instruction of the far loop (whose v_rsq_f32 come in bursts of 16); behind any smaller set — one class, one region, fewer idle cycles — the fault stays;
      * the same source compiled so that its near loop has no packed instruction (-fno-slp-vectorize: every packed instruction of the far loops is written as one in the
        source) runs clean with or without the fence.
    So what is excluded is the combination the fault needs: a kernel whose waves issue BURST or more transcendental instructions back to back (matrix-core and scalar
    instructions between them not counted) must not have a v_pk_*_f32 in any loop that evaluates masked exact pairs (a loop with a transcendental instruction, a
    v_cndmask and no matrix-core instruction).
RULE B (fails the check) — no inline-asm body reads a register that a transcendental or matrix-core instruction wrote within WINDOW issue slots unless a compiler-made
    instruction read it first: the compiler pads the hazards it knows between its own instructions and cannot see into an asm body
    (include/sctl_amd/device/ukernels.hpp: rsqrt_masked, where an asm block read a stale v_rsq_f64 result).  Asm bodies are what hipcc brackets with ;APP / ;NO_APP.
LISTED, not a failure — vector instructions that overwrite a transcendental's SOURCE register within two issue slots, before its result is read.  The first suspect of
    round 3 (`v_rsq_f32 v34, v36 ; v_pk_mul_f32 v[36:37], ...`), REFUTED in round 4: with that instruction's destination renamed to a free register, or 1 to 32 wait
    states between the two, the kernel is as faulty as before (tools/kernel_repeat_war.py, profiles/r04_kernel_repeat_war.txt); every kernel with a seed-only
    reciprocal square root has such pairs.

    python tools/check_isa_rules.py [--list] file.s ...     exit code 1 on a finding of rule A or B
    python tools/check_isa_rules.py --shipped [--list]     compiles every translation unit of sctl_amd/csrc and the test plugin with the Makefile's flags and checks them
"""
import collections, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BURST = 12           # transcendental instructions back to back: the kernels the fault was seen with issue 16, the exact kernels and the plugin at most 9
WINDOW = 16          # issue slots behind a transcendental / matrix-core instruction within which an asm body may not be the first reader of its result
TRANS = ('v_rsq_', 'v_rcp_', 'v_sqrt_', 'v_exp_', 'v_log_', 'v_sin_', 'v_cos_')
LOADS = ('ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle', 'global_load', 'buffer_load', 'flat_load', 'scratch_load')
NO_VDST = ('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')


def vregs(tok):
    """vector registers named by one operand (modifiers such as -v[2:3] or |v5| stripped)"""
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(line):
    """(mnemonic, vector registers written, vector registers read, issue slots) of an instruction line, else None"""
    t = line.strip()
    if not line.startswith('\t') or not t or t[0] in '.;':
        return None
    t = t.split(';')[0].strip()
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    wr, rd = set(), set()
    if op.startswith('v_') and not op.startswith(NO_VDST) or op.startswith(LOADS):
        wr = vregs(ops[0]) if ops else set()
        for o in ops[1:]:
            rd |= vregs(o)
        if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot2c', 'v_dot4c', 'v_dot8c')):
            rd |= wr                                    # the destination is an operand too
        if op.startswith('v_swap'):
            wr |= vregs(ops[1]); rd |= vregs(ops[0])
    else:
        for o in ops:
            rd |= vregs(o)
    slots = 1
    if op == 's_nop':
        slots = int(ops[0], 0) + 1
    elif op.startswith('s_waitcnt'):
        slots = 0
    return op, wr, rd, slots


def functions(src):
    """(name, lines) of every function of an assembly file"""
    for m in re.finditer(r'^(_Z\w+):', src, re.M):
        end = src.find('.Lfunc_end', m.end())
        if end > 0:
            yield m.group(1), src[m.start():end].split('\n')


def longest_burst(ins):
    run = best = 0
    for p in ins:
        if p is None:
            continue
        op = p[0]
        if op.startswith(TRANS):
            run += 1
            best = max(best, run)
        elif not op.startswith(('v_mfma', 's_')):
            run = 0
    return best


def loops(body, ins):
    """(first line, last line, mnemonic counts) of every backward branch's span"""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    for i, p in enumerate(ins):
        if p and p[0].startswith('s_cbranch'):
            tgt = body[i].split()[-1]
            if tgt in labels and labels[tgt] < i:
                yield labels[tgt], i, collections.Counter(q[0] for q in ins[labels[tgt]:i + 1] if q)


def check(body, window=WINDOW):
    """-> (longest transcendental burst, rule A findings, rule B findings, listed source overwrites)"""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    ins = [parse(l) for l in body]
    in_asm, flag = [], False
    for l in body:
        s = l.strip()
        if s.startswith(';APP'):
            flag = True
        elif s.startswith(';NO_APP'):
            flag = False
        in_asm.append(flag)
    # ---- rule A -------------------------------------------------------------------------------------------------------------------------
    burst = longest_burst(ins)
    rule_a = []
    if burst >= BURST:
        for lo, hi, c in loops(body, ins):
            n = lambda pre, suf='': sum(v for k, v in c.items() if k.startswith(pre) and suf in k)
            if n(TRANS) and n('v_cndmask') and not n('v_mfma') and n('v_pk_', 'f32'):
                first = next(i for i in range(lo, hi + 1) if ins[i] and ins[i][0].startswith('v_pk_') and 'f32' in ins[i][0])
                rule_a.append((lo, hi, n('v_pk_', 'f32'), first, body[first].strip()))
    # ---- rule B and the listing ------------------------------------------------------------------------------------------------------------
    rule_b, listed = [], []

    def walk(i, left, src, dst, origin, seen):
        while left > 0 and i < len(body):
            if (i, left) in seen:
                return
            seen.add((i, left))
            p = ins[i]
            if p is None:
                i += 1
                continue
            op, wr, rd, slots = p
            if op.startswith(('s_cbranch', 's_branch')):
                tgt = body[i].split()[-1]
                if tgt in labels:
                    walk(labels[tgt], left - 1, src, dst, origin, seen)
                if op.startswith('s_branch'):
                    return
                left -= 1
                i += 1
                continue
            if op.startswith(('s_endpgm', 's_setpc', 's_swappc')):
                return
            if rd & dst:
                if in_asm[i]:
                    rule_b.append((origin, body[origin].strip(), i, body[i].strip()))
                return                                   # the first reader of the result
            if wr & src and window - left < 2 and op.startswith('v_') and not op.startswith(TRANS + ('v_mfma',)):
                listed.append((origin, body[origin].strip(), i, body[i].strip()))
                src = set()
            if wr & dst:
                return                                   # the result itself is overwritten
            left -= slots
            i += 1

    ntrans = 0
    for i, p in enumerate(ins):
        if p is None or in_asm[i]:
            continue
        op, wr, rd, _ = p
        if op.startswith(TRANS):
            ntrans += 1
            walk(i + 1, window, set() if wr & rd else rd, wr, i, set())
        elif op.startswith('v_mfma'):
            walk(i + 1, window, set(), wr, i, set())
    return ntrans, burst, rule_a, rule_b, listed


def check_file(path, verbose=False, out=sys.stdout):
    src = open(path).read()
    bad = nfun = ntrans = nlisted = nburst = 0
    for name, body in functions(src):
        n, burst, ra, rb, listed = check(body)
        nfun += 1
        ntrans += n
        nlisted += len(listed)
        nburst += burst >= BURST
        if ra or rb:
            bad += len(ra) + len(rb)
            print("%s: %s: bursts of %d transcendental instructions; rule A: %d masked-pair loop(s) with packed fp32 instructions; rule B: %d asm read(s) of a fresh "
                  "transcendental / matrix-core result" % (os.path.basename(path), name, burst, len(ra), len(rb)), file=out)
            for lo, hi, npk, first, text in ra[:8]:
                print("      A: loop at lines %d-%d: %d v_pk_*_f32, the first at line %d  %s" % (lo, hi, npk, first, text), file=out)
            for o, ol, i, il in rb[:8]:
                print("      B: line %d  %s   <- asm, line %d  %s" % (o, ol, i, il), file=out)
        if verbose and listed:
            print("%s: %s: %d transcendental source(s) overwritten within two slots (listed)" % (os.path.basename(path), name, len(listed)), file=out)
            for o, ol, i, il in listed[:4]:
                print("      line %d  %s ; line %d  %s" % (o, ol, i, il), file=out)
    print("%s: %d functions (%d with transcendental bursts >= %d), %d transcendental instructions, %d finding(s); listed source overwrites: %d"
          % (os.path.basename(path), nfun, nburst, BURST, ntrans, bad, nlisted), file=out)
    return bad


def shipped_sources():
    csrc = os.path.join(ROOT, "sctl_amd", "csrc")
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.hip')) + [os.path.join(ROOT, "tests", "plugin", "yukawa_kernel.hip")]


def compile_asm(srcs, outdir, extra=()):
    """device assembly of each source with the flags the Makefile gives its translation unit"""
    mk = lambda *a: subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sctl_amd", "csrc")] + list(a), capture_output=True, text=True).stdout.split()
    flags = mk("print-flags")

    def one(s):
        o = os.path.join(outdir, os.path.basename(s)[:-4] + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + mk("print-unit-flags", "UNIT=" + os.path.basename(s)[:-4]) + list(extra) + ["--offload-device-only", "-S", s, "-o", o],
                       check=True, stderr=subprocess.DEVNULL)
        return o
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return list(ex.map(one, srcs))


def main():
    args = sys.argv[1:]
    verbose = '--list' in args
    args = [a for a in args if a != '--list']
    bad = 0
    if '--shipped' in args:
        with tempfile.TemporaryDirectory() as td:
            for f in compile_asm(shipped_sources(), td):
                bad += check_file(f, verbose)
    for f in args:
        if f != '--shipped':
            bad += check_file(f, verbose)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
