#!/usr/bin/env python3
"""A/B of the plain (T) wall time between builds of libafesp_hip.so in ONE GPU session.
usage: ab_triples.py libA.so libB.so [...] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
out = []
for (o, v, reps) in ((20, 200, 4), (5, 53, 200)):
    eng.synthetic_init(o, v, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
    e = eng.do_ccsd_t_spatial_plain()
    t0 = time.perf_counter()
    for r in range(reps): e = eng.do_ccsd_t_spatial_plain()
    out.append("o=%%d v=%%d plain %%.3f ms  E=%%.12e" %% (o, v, (time.perf_counter() - t0) / reps * 1e3, e[0]))
    e = eng.do_ccsd_t_spatial()
    t0 = time.perf_counter()
    for r in range(reps): e = eng.do_ccsd_t_spatial()
    out.append("full %%.3f ms" %% ((time.perf_counter() - t0) / reps * 1e3))
print("  ".join(out))
eng.close()
''' % ROOT
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2
for r in range(rounds):
    for tag, lib in zip("ABCDEFGH", libs):
        env = dict(os.environ, AFESP_LIBRARY=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(tag, res.stdout.strip() or res.stderr.strip()[-300:], flush=True)
