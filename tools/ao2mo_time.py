#!/usr/bin/env python3
"""AO->MO transform (+ repack + MP2) on resident synthetic AO integrals: wall time per call; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = o + v
q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
eng = Engine(0)
eng.synthetic_ao(n, 0.02, 777)
for r in range(reps + 1):
    t0 = time.perf_counter()
    emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)
    print("call", r, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), "E(MP2)", emp2, flush=True)
eng.close()
