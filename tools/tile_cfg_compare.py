#!/usr/bin/env python3
"""Forced tile configurations on the (T)-shaped and the long-K products, one session."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
M, N = 40000, 8192
for rnd in range(2):
    for (tm, tn, name) in ((0, 0, "auto (256x128, 8 waves)"), (8, 8, "128x128, 4 waves (2 workgroups/CU)"), (4, 4, "128x128, 8 waves"), (8, 16, "128x256, 8 waves")):
        out = []
        for K in (224, 3520):
            eng.set_tuning(0, tm, tn, 0)
            ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=5)
            out.append("K=%d %.2f TF" % (K, 2.0 * M * N * K / ms / 1e9))
        print(name, "  ".join(out), flush=True)
eng.set_tuning(0, 0, 0, 0)
eng.close()
