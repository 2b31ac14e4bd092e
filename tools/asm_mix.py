"""Instruction mix per basic block of one gett_kernel instantiation (device asm from hipcc -S)."""
import re, subprocess, sys
from collections import Counter
import os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a-fortran-electronic-structure-program_amd", "csrc", "gett.hip")
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only", src, "-o", "/tmp/gett.s"],
               stderr=subprocess.DEVNULL, check=True)
s = open('/tmp/gett.s').read()
name = sys.argv[1] if len(sys.argv) > 1 else '_ZN5afesp11gett_kernelILi4ELi2ELi4ELi4ELb1ELb1ELi2EEEvNS_14GettKernelArgsE'
i = s.index(name + ':')
body = s[i:s.index('s_endpgm', i)]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.set')]
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in lines:
    if re.match(r'^\.?[A-Za-z_0-9$]+:', l):
        cur = [l[:12], []]; blocks.append(cur)
    else:
        cur[1].append(l)
def cls(op):
    if 'mfma' in op: return 'mfma'
    for p, k in (('ds_read', 'ds_read'), ('ds_write', 'ds_write'), ('global_load', 'gload'), ('global_store', 'gstore'), ('scratch', 'scratch'),
                 ('v_accvgpr', 'accvgpr'), ('v_', 'valu'), ('s_waitcnt', 'waitcnt'), ('s_barrier', 'barrier'), ('s_nop', 's_nop'),
                 ('s_cbranch', 'branch'), ('s_branch', 'branch'), ('s_', 'salu')):
        if op.startswith(p): return k
    return op
for b in blocks:
    c = Counter(cls(x.split()[0]) for x in b[1])
    if c.get('mfma', 0) >= 1: print(b[0], len(b[1]), dict(c))
m = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s[s.index('amdhsa.kernels'):], re.S)
if m: print("vgpr", m.group(1), "spill", m.group(2))
