#!/usr/bin/env python3
"""A/B of two builds of libafesp_hip.so in ONE GPU session (devices differ by several per cent, so numbers from
different gpurun calls are not comparable).  usage: ab_gemm.py libA.so libB.so [libC.so ...] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 20, 200
out = []
M, N = 40000, 8192
for K in (224, 3520):
    ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=5)
    out.append("K=%%d %%.2f TF" %% (K, 2.0 * M * N * K / ms / 1e9))
ms = eng.bench_contract("mjae", (o, o, v, v), "iemb", (o, v, o, v), "ijab", (o, o, v, v), reps=5)
out.append("ring %%.2f TF" %% (2.0 * o**3 * v**3 / ms / 1e9))
ms = eng.bench_contract("ijef", (o, o, v, v), "efab", (v, v, v, v), "ijab", (o, o, v, v), reps=3)
out.append("ladder %%.2f TF" %% (2.0 * o**2 * v**4 / ms / 1e9))
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
