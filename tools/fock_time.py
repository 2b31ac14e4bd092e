#!/usr/bin/env python3
"""afesp_build_fock (src/hf.f90:349-385) on resident synthetic AO integrals: wall time per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(3)
d = rng.standard_normal((n, n)); d = d + d.T
h = rng.standard_normal((n, n)); h = h + h.T
eng = Engine(0)
eng.synthetic_ao(n, 0.02, 777)
for r in range(reps + 1):
    t0 = time.perf_counter()
    f = eng.build_fock(n, d, h)
    print("call", r, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), "checksum %.12e" % float(np.sum(f * d)), flush=True)
eng.close()
