#!/usr/bin/env python3
"""Compact view of the blocks of one gett_kernel instantiation that contain MFMAs, LDS writes or barriers.
usage: loop_dump.py /tmp/gett_vN.s [mangled-name-substring]"""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else "Li4ELi2ELi4ELi4ELb1ELb1ELi2ELb1ELb0EE"
m = re.search(r"^(_ZN5afesp11gett_kernelI" + re.escape(sub) + r"[^:\n]*):", s, re.M)
i = m.start()
body = s[i:s.index('s_endpgm', i)]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in lines:
    if re.match(r'^\.?[A-Za-z_0-9$]+:', l): cur = [l.split(':')[0], []]; blocks.append(cur)
    else: cur[1].append(l)
KEEP = ('ds_write', 's_barrier', 's_cbranch', 's_branch', 'global_load', 'global_store', 's_waitcnt', 'scratch', 's_setprio')
for b in blocks:
    txt = ' '.join(b[1])
    if not ('mfma' in txt or 'ds_write' in txt or 's_barrier' in txt): continue
    out = []; n = 0; k = 0; v = 0
    def flush():
        global n, k, v
        if n: out.append(f"[{n} mfma]"); n = 0
        if k: out.append(f"[{k} ds_read]"); k = 0
        if v: out.append(f"[{v} valu]"); v = 0
    for l in b[1]:
        op = l.split()[0]
        if 'mfma' in op:
            if k or v: flush()
            n += 1
        elif op.startswith('ds_read'):
            if n or v: flush()
            k += 1
        elif op.startswith('v_'):
            if n or k: flush()
            v += 1
        elif op.startswith(KEEP):
            flush(); out.append(' '.join(l.split(';')[0].split()[:2])[:28])
    flush()
    print(b[0], '|', ' | '.join(out)[:1500]); print()
