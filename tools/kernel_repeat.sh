#!/bin/bash
# Builds code objects of sctl_amd/csrc/centered.hip and runs tools/ubench/kernel_repeat on the matrix-core kernels in them:
#   shipped          the source as it is (near pairs fenced one after the other)
#   nofence          -DSCTL_AMD_EXP_NO_NEAR_FENCE: the form that gave run-to-run different near sums
#   nofence_pairnop  nofence's ASSEMBLY with one `s_nop 0` between every `v_rsq_f32 vA, vB` and a directly following v_pk_* that overwrites vB
#   nofence_allnop   nofence's assembly with `s_nop 3` behind every instruction of the double-layer / 128-target kernel
#   nofence_nearnop / nofence_farnop   the same nops only inside / only outside the near-flush region of that kernel
# part 1 (no GPU): tools/kernel_repeat.sh build      part 2 (GPU box): tools/kernel_repeat.sh run
set -e
cd "$(dirname "$0")/.."
D=tools/ab/repeat
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="$(make -s -C sctl_amd/csrc print-flags) $(make -s -C sctl_amd/csrc print-unit-flags UNIT=centered) --offload-device-only -S"
co() { $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/$1.s -o $D/$1.o && $LLVM/ld.lld -shared $D/$1.o -o $D/$1.co && rm -f $D/$1.o; }
if [ "$1" = build ]; then
  mkdir -p $D
  /opt/rocm/bin/hipcc $FLAGS sctl_amd/csrc/centered.hip -o $D/shipped.s 2>/dev/null; co shipped
  /opt/rocm/bin/hipcc $FLAGS -DSCTL_AMD_EXPERIMENTS -DSCTL_AMD_EXP_NO_NEAR_FENCE sctl_amd/csrc/centered.hip -o $D/nofence.s 2>/dev/null; co nofence
  python3 - <<'PY'
import re
D='tools/ab/repeat/'
src=open(D+'nofence.s').read()
sym='_ZN8sctl_amd24centered_mfma_f32_kernelILb1ELi4EEEvNS_8EvalArgsIfEE'
i0=src.index('\n'+sym+':'); i1=src.index('.Lfunc_end',i0)
body=src[i0:i1].split('\n')
def regs(tok):
    tok=tok.strip()
    m=re.match(r'v\[(\d+):(\d+)\]$',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    return {int(m.group(1))} if m else set()
isins=lambda l: l.startswith('\t') and l.strip() and l.strip()[0] not in '.;'
idx=[i for i,l in enumerate(body) if isins(l)]
nxt={idx[k]:idx[k+1] for k in range(len(idx)-1)}
lab={m.group(1):i for i,l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):',l)] if m}
loops=[]
for i,l in enumerate(body):
    m=re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in lab and lab[m.group(1)]<i:
        seg=body[lab[m.group(1)]:i+1]
        if any('v_cndmask' in x for x in seg) and any('v_rsq' in x for x in seg) and not any('v_mfma' in x for x in seg): loops.append((lab[m.group(1)],i))
A,B=min(x for x,_ in loops[:3])-60,max(y for _,y in loops[:3])+80      # the near flush between tiles, with its entry and exit
pair=set()
for i in idx[:-1]:
    t=body[i].strip()
    if t.startswith('v_rsq_f32'):
        t2=body[nxt[i]].strip()
        if t2.startswith('v_pk_') and regs(t2.split(None,1)[1].split(',')[0]) & regs(t.split(None,1)[1].split(',')[1]): pair.add(i)
def write(name, after, nop):
    out=[]; n=0
    for i,l in enumerate(body):
        out.append(l); t=l.strip()
        if isins(l) and not t.startswith(('s_cbranch','s_branch','s_endpgm','s_setpc','s_waitcnt')) and after(i): out.append('\t'+nop); n+=1
    open(D+name+'.s','w').write(src[:i0]+'\n'.join(out)+src[i1:]); print(name, n, 'x', nop)
write('nofence_pairnop', lambda i: i in pair, 's_nop 0')
write('nofence_allnop', lambda i: True, 's_nop 3')
write('nofence_nearnop', lambda i: A<=i<=B, 's_nop 3')
write('nofence_farnop', lambda i: not (A<=i<=B), 's_nop 3')
PY
  for v in nofence_pairnop nofence_allnop nofence_nearnop nofence_farnop; do co $v; done
  rm -f $D/*.s
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/kernel_repeat.cpp -o tools/ubench/kernel_repeat
  ls $D
else
  echo "# tools/kernel_repeat.sh run: 24 launches of one kernel on one problem (2^17 targets x 2^16 sources, fp32), ANOTHER KERNEL between any two launches (POISON=1);"
  echo "# a run counts as off when any of its values differs from the value most runs give"
  for k in ILb1ELi4E ILb1ELi8E ILb0ELi4E fxu256; do for v in shipped nofence; do KERNEL=$k POISON=1 tools/ubench/kernel_repeat $D/$v.co; done; done
  echo "# the double-layer / 128-target kernel of the no-fence build, its assembly patched with idle instructions"
  for v in nofence_pairnop nofence_allnop nofence_nearnop nofence_farnop; do POISON=1 tools/ubench/kernel_repeat $D/$v.co; done
  echo "# the no-fence kernel, launches back to back without another kernel in between, and with LDS / vector registers filled with NaN patterns in between"
  tools/ubench/kernel_repeat $D/nofence.co; POISON=0x7fc00000 tools/ubench/kernel_repeat $D/nofence.co; POISON_VGPR=1 tools/ubench/kernel_repeat $D/nofence.co
fi
