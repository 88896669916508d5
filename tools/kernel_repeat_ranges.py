"""Bisection aid for tools/kernel_repeat.sh: code objects of the no-fence build in which `s_nop 3` follows every instruction whose index
(counting the instructions of centered_mfma_f32_kernel<true, 4>) lies in a given range.
    python tools/kernel_repeat_ranges.py 678:1130 904:1130 ...   ->  tools/ab/repeat/r_<lo>_<hi>.co, to be run with POISON=1 tools/ubench/kernel_repeat"""
import re,sys,subprocess
D='tools/ab/repeat/'
# nofence.s was removed by the build step; regenerate it if needed
import os
if not os.path.exists(D+'nofence.s'):
    flags=subprocess.run(['make','-s','-C','sctl_amd/csrc','print-flags'],capture_output=True,text=True).stdout.split()
    subprocess.run(['/opt/rocm/bin/hipcc']+flags+['--offload-device-only','-S','-DSCTL_AMD_EXPERIMENTS','-DSCTL_AMD_EXP_NO_NEAR_FENCE','sctl_amd/csrc/centered.hip','-o',D+'nofence.s'],check=True,stderr=subprocess.DEVNULL)
src=open(D+'nofence.s').read()
sym='_ZN8sctl_amd24centered_mfma_f32_kernelILb1ELi4EEEvNS_8EvalArgsIfEE'
i0=src.index('\n'+sym+':'); i1=src.index('.Lfunc_end',i0)
body=src[i0:i1].split('\n')
isins=lambda l: l.startswith('\t') and l.strip() and l.strip()[0] not in '.;' and not l.strip().startswith(('s_cbranch','s_branch','s_endpgm','s_setpc','s_waitcnt'))
idx=[i for i,l in enumerate(body) if isins(l)]
print('instructions',len(idx))
def write(name, lo, hi):
    sel=set(idx[lo:hi]); out=[]
    for i,l in enumerate(body):
        out.append(l)
        if i in sel: out.append('\ts_nop 3')
    open(D+name+'.s','w').write(src[:i0]+'\n'.join(out)+src[i1:])
    subprocess.run(['/opt/rocm/lib/llvm/bin/clang','-x','assembler','-target','amdgcn-amd-amdhsa','-mcpu=gfx950','-c',D+name+'.s','-o',D+name+'.o'],check=True)
    subprocess.run(['/opt/rocm/lib/llvm/bin/ld.lld','-shared',D+name+'.o','-o',D+name+'.co'],check=True)
    os.remove(D+name+'.o'); os.remove(D+name+'.s')
for spec in sys.argv[1:]:
    lo,hi=map(int,spec.split(':'))
    write('r_%d_%d'%(lo,hi),lo,hi)
