#!/usr/bin/env python3
"""A/B of small-system timings (CCSD iteration, (T)) and of mid-size contractions on the 4-wave tiles between builds of libafesp_hip.so in ONE GPU
session.  usage: ab_small.py libA.so libB.so [...] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
out = []
for (o, v) in ((5, 53), (7, 21), (10, 100)):
    eng.synthetic_init(o, v, 0.02, 12345, 8)
    eng.ccsd_energy()
    for _ in range(5): eng.ccsd_iterate(); eng.ccsd_diis()
    reps = 100 if o * v < 600 else 20
    t0 = time.perf_counter()
    for _ in range(reps): eng.ccsd_iterate(); eng.ccsd_diis()
    it = (time.perf_counter() - t0) / reps
    eng.do_ccsd_t_spatial_plain()
    t0 = time.perf_counter()
    for _ in range(reps): e = eng.do_ccsd_t_spatial_plain()
    tt = (time.perf_counter() - t0) / reps
    out.append("o=%%d v=%%d iter %%.3f ms (T) %%.3f ms" %% (o, v, it * 1e3, tt * 1e3))
# skinny products on the 4-wave tiles: N = 64 columns, long K
for (M, N, K) in ((8192, 64, 4096), (40000, 64, 512), (2048, 32, 8192)):
    ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=20)
    out.append("%%dx%%dx%%d %%.2f TF" %% (M, N, K, 2.0 * M * N * K / ms / 1e9))
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
