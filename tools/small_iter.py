#!/usr/bin/env python3
"""Small-system CCSD iteration and (T) times, laned launches (AFESP_NO_GRAPH=1) and replayed graph, for the current environment
(tuning knobs).  usage: small_iter.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("AFESP_PKG", os.path.join(ROOT, "a-fortran-electronic-structure-program_amd")))
from afesp_amd.capi import Engine
out = []
with Engine(0) as eng:
    for (o, v) in ((5, 53), (7, 21), (9, 19)):
        eng.synthetic_init(o, v, 0.02, 12345, 8)
        eng.ccsd_energy()
        for _ in range(5): eng.ccsd_iterate(); eng.ccsd_diis()
        t0 = time.perf_counter()
        for _ in range(30): eng.ccsd_iterate(); eng.ccsd_diis()
        it = (time.perf_counter() - t0) / 30
        for _ in range(20): eng.ccsd_iterate(); eng.ccsd_diis()     # (past the 40 calls after which the graph is captured)
        t0 = time.perf_counter()
        for _ in range(50): eng.ccsd_iterate(); eng.ccsd_diis()
        rp = (time.perf_counter() - t0) / 50
        eng.do_ccsd_t_spatial_plain()
        t0 = time.perf_counter()
        for _ in range(50): eng.do_ccsd_t_spatial_plain()
        tt = (time.perf_counter() - t0) / 50
        out.append("o=%d v=%d iter %.3f ms laned / %.3f ms replayed, (T) %.3f ms" % (o, v, it * 1e3, rp * 1e3, tt * 1e3))
print(" | ".join(out), flush=True)
