"""Bisection aid for the run-to-run different near sums of the no-fence build: code objects of that build in which idle instructions follow
selected instructions of centered_mfma_f32_kernel<true, 4>.  A spec is a '+'-joined list of
    lo:hi[:prefix] the instructions with index in [lo, hi) [whose mnemonic starts with the prefix] (counting the kernel's instructions, branches and s_waitcnt excluded: 1807 at round 4)
    class:<prefix>[-<prefix>...] every instruction whose mnemonic starts with the first prefix and with none of the others (class:ds_read, class:v_-v_pk_)
optionally followed by @k: the idle instruction is `s_nop k` (default 3).
    python tools/kernel_repeat_ranges.py build [--from-source] 750:820+1090:1130 class:v_mfma@7 ...   ->  tools/ab/repeat/r_<spec>.co   (no GPU; the kernel is
                                                     profiles/r04_near_fault/kernel_nofence.s, or with --from-source today's no-fence build of centered.hip)
    python tools/kernel_repeat_ranges.py run                                            ->  every tools/ab/repeat/r_*.co through tools/ubench/kernel_repeat, POISON=1 (GPU box)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, 'tools', 'ab', 'repeat') + '/'
SYM = '_ZN8sctl_amd24centered_mfma_f32_kernelILb1ELi4EEEvNS_8EvalArgsIfEE'
SKIP = ('s_cbranch', 's_branch', 's_endpgm', 's_setpc')


FROZEN = os.path.join(ROOT, 'profiles', 'r04_near_fault', 'kernel_nofence.s')   # that kernel alone, as round 4's compiler made it: the reproducer


def source(from_source=False):
    """the assembly the variants are cut from: the frozen reproducer, or (--from-source) today's no-fence, SLP-vectorised build of centered.hip"""
    os.makedirs(D, exist_ok=True)
    if not from_source:
        return open(FROZEN).read()
    mk = lambda *a: subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'sctl_amd', 'csrc')] + list(a), capture_output=True, text=True).stdout.split()
    flags = mk('print-flags') + [f for f in mk('print-unit-flags', 'UNIT=centered') if f != '-fno-slp-vectorize']
    s = D + 'nofence_src.s'
    subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['--offload-device-only', '-S', '-DSCTL_AMD_EXPERIMENTS', '-DSCTL_AMD_EXP_NO_NEAR_FENCE',
                    os.path.join(ROOT, 'sctl_amd', 'csrc', 'centered.hip'), '-o', s], check=True, stderr=subprocess.DEVNULL)
    src = open(s).read()
    os.remove(s)
    return src


def build(specs):
    from_source = '--from-source' in specs
    specs = [x for x in specs if x != '--from-source']
    src = source(from_source)
    i0 = src.index('\n' + SYM + ':') + 1
    i1 = src.index('.Lfunc_end', i0)
    body = src[i0:i1].split('\n')
    isany = lambda l: l.startswith('\t') and l.strip() and l.strip()[0] not in '.;' and not l.strip().startswith(SKIP)
    isins = lambda l: isany(l) and not l.strip().startswith('s_waitcnt')
    idx = [i for i, l in enumerate(body) if isins(l)]
    print('instructions', len(idx))
    for spec in ['plain'] + specs:   # 'plain': the reproducer unchanged
        sel, nop = set(), 3
        if spec == 'plain':
            spec = '0:0'
        s = spec
        if '@' in s:
            s, k = s.split('@'); nop = int(k)
        for part in s.split('+'):
            if part.startswith('class:'):   # class:v_-v_pk_-v_mov = every v_ instruction that is neither v_pk_ nor v_mov
                inc, *exc = part[6:].split('-')
                sel.update(i for i, l in enumerate(body) if isany(l) and l.strip().startswith(inc) and not l.strip().startswith(tuple(exc) or ('\0',)))
            elif part.count(':') == 2:   # lo:hi:prefix — the instructions of that index range whose mnemonic starts with the prefix
                lo, hi, pre = part.split(':')
                sel.update(i for i in idx[int(lo):int(hi)] if body[i].strip().startswith(pre))
            else:
                lo, hi = map(int, part.split(':'))
                sel.update(idx[lo:hi])
        out = []
        for i, l in enumerate(body):
            out.append(l)
            if i in sel:
                out.append('\ts_nop %d' % nop)
        name = 'r_' + re.sub(r'[^0-9A-Za-z_]+', '_', spec)
        open(D + name + '.s', 'w').write(src[:i0] + '\n'.join(out) + src[i1:])
        subprocess.run(['/opt/rocm/lib/llvm/bin/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', D + name + '.s', '-o', D + name + '.o'], check=True)
        subprocess.run(['/opt/rocm/lib/llvm/bin/ld.lld', '-shared', D + name + '.o', '-o', D + name + '.co'], check=True)
        os.remove(D + name + '.o'); os.remove(D + name + '.s')
        print('%-44s %4d x s_nop %d' % (name, len(sel), nop))
    exe = os.path.join(ROOT, 'tools', 'ubench', 'kernel_repeat')
    if not os.path.exists(exe):
        subprocess.run(['/opt/rocm/bin/hipcc', '-O2', '--offload-arch=gfx950', exe + '.cpp', '-o', exe], check=True)


def run():
    env = dict(os.environ, POISON='1')
    for co in sorted(glob.glob(D + 'r_*.co')):
        subprocess.run([os.path.join(ROOT, 'tools', 'ubench', 'kernel_repeat'), os.path.relpath(co, ROOT)], env=env, cwd=ROOT, check=True)
        sys.stdout.flush()


if __name__ == '__main__':
    if sys.argv[1:2] == ['build']:
        build(sys.argv[2:])
    else:
        run()
