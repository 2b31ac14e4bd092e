#!/usr/bin/env python3
"""AO->MO + MP2 of an H2O/cc-pVTZ-shaped system (n = 58): wall time per call with the one-launch MP2 energy and with the five-launch
form (AFESP_MP2_PACKED=0), alternating in one process."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = 5, 53
n = o + v
q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(v) / (v - 1)])
with Engine(0) as eng:
    eng.synthetic_ao(n, 0.02, 777)
    for rep in range(3):
        for mode in ("1", "0"):
            os.environ[os.environ.get("AB_KNOB", "AFESP_MP2_PACKED")] = mode
            for _ in range(5): emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)
            ts = []
            for _ in range(41):
                t0 = time.perf_counter(); emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False); ts.append(time.perf_counter() - t0)
            print(f"{os.environ.get('AB_KNOB', 'AFESP_MP2_PACKED')}={mode}: median {np.median(ts)*1e6:7.1f} us  min {min(ts)*1e6:7.1f} us  E(MP2) {emp2:.12f}", flush=True)
