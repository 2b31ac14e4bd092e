#!/usr/bin/env python3
"""AO->MO + MP2 at n = 220 (config 5): wall time per call with a knob (AB_KNOB, default AFESP_PSQ_ORDER) alternating between 1 and 0
in one process, E(MP2) of both printed (must be identical)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
n = o + v
knob = os.environ.get("AB_KNOB", "AFESP_PSQ_ORDER")
q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
with Engine(0) as eng:
    eng.synthetic_ao(n, 0.02, 777)
    for rep in range(3):
        for mode in ("1", "0"):
            os.environ[knob] = mode
            eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False); ts.append(time.perf_counter() - t0)
            print(f"{knob}={mode}: median {np.median(ts)*1e3:7.2f} ms  min {min(ts)*1e3:7.2f} ms  E(MP2) {emp2:.13f}", flush=True)
