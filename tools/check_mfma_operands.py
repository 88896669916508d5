#!/usr/bin/env python3
"""Static check of the device assembly (no GPU needed) for an instruction pattern the compiler's hazard rules do not exclude and this repository keeps
out of its matrix-core kernels (sctl_amd/csrc/centered_mfma_kernel.hpp): a VALU / LDS / memory-load write to a register that a recently
issued v_mfma reads as its A or B operand.  A precaution — whether such a write can land before an MFMA that waits in the matrix pipe has read the register
is not documented —: for every v_mfma of the matrix-core kernels the next WINDOW vector / LDS / memory instructions along both arms of every branch must not
write the MFMA's A / B registers (its own destination included).
(What used to be listed here — vector instructions overwriting a transcendental's source — is tools/check_isa_rules.py's now, with what round 4 found about it.)
    python tools/check_mfma_operands.py [asm file]      exit code 1 on a finding; without a file it compiles sctl_amd/csrc/centered.hip"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 24          # instructions that occupy the vector issue (>= 4 cycles each: ~100 cycles, three MFMA slots of 32)


def regs(tok):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def parse(line):
    """(mnemonic, written vector registers, operands) of one instruction line, or None"""
    t = line.strip()
    if not line.startswith('\t') or not t or t[0] in '.;':
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    dst = set()
    if op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
        dst = set()
    elif op.startswith('v_') or op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        dst = regs(ops[0]) if ops else set()
        if op.startswith('v_swap') or op.startswith('v_permlane') and 'swap' in op:
            dst |= regs(ops[1])
    return op, dst, ops


def kernels(src, token):
    for m in re.finditer(r'^(_Z\w*%s\w*):' % token, src, re.M):
        i0 = m.start()
        yield m.group(1), src[i0:src.index('.Lfunc_end', i0)].split('\n')


def check(body):
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    found = []

    def walk(i, left, watch, origin, seen):
        while left > 0 and i < len(body):
            if (i, left) in seen:
                return
            seen.add((i, left))
            p = parse(body[i])
            if p is None:
                i += 1
                continue
            op, dst, ops = p
            if op.startswith(('s_cbranch', 's_branch')) and ops and ops[0] in labels:
                walk(labels[ops[0]], left, watch, origin, seen)
                if op.startswith('s_branch'):
                    return
                i += 1
                continue
            if op.startswith('s_endpgm'):
                return
            if op.startswith(('v_', 'ds_', 'global_', 'buffer_', 'flat_')):
                if dst & watch:
                    found.append((origin, body[origin].strip(), i, body[i].strip()))
                left -= 1
            i += 1

    n = 0
    for i, l in enumerate(body):
        p = parse(l)
        if p and p[0].startswith('v_mfma'):
            n += 1
            op, dst, ops = p
            ab = regs(ops[1]) | regs(ops[2])
            if dst & ab:
                found.append((i, l.strip(), i, "destination overlaps its own A / B operand"))
            walk(i + 1, WINDOW, ab, i, set())
    return n, found


def main():
    if len(sys.argv) > 1:
        src = open(sys.argv[1]).read()
    else:
        flags = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sctl_amd", "csrc"), "print-flags"], capture_output=True, text=True).stdout.split()
        flags += subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sctl_amd", "csrc"), "print-unit-flags", "UNIT=centered"], capture_output=True, text=True).stdout.split()
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "centered.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--offload-device-only", "-S", os.path.join(ROOT, "sctl_amd", "csrc", "centered.hip"), "-o", asm],
                           check=True, stderr=subprocess.DEVNULL)
            src = open(asm).read()
    bad = 0
    for name, body in kernels(src, "centered_mfma"):
        n, found = check(body)
        print("%s: %d v_mfma, %d operand write(s) within %d instructions" % (name, n, len(found), WINDOW))
        for o, ol, i, il in found[:10]:
            print("   line %d  %s\n      <- line %d  %s" % (o, ol, i, il))
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
