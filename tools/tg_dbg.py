#!/usr/bin/env python3
"""Grouped (T) GEMM time per evaluation at config-5 extents (or `o v`) from the engine's HIP events; run once per AFESP_TG_DBG /
AFESP_T_GEMM setting.  usage: tg_dbg.py [o v [reps]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
with Engine(0) as eng:
    eng.synthetic_init(o, v, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
    eng.do_ccsd_t_spatial_plain()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps): out = eng.do_ccsd_t_spatial_plain()
    dt = (time.perf_counter() - t0) / reps
    p = eng.profile(False)
    print("dbg=%s gemm=%s: (T) %.2f ms, GEMM %.2f ms = %.2f TF, orbit %.2f ms" % (os.environ.get("AFESP_TG_DBG", "0"), os.environ.get("AFESP_T_GEMM", "tg"),
          dt * 1e3, p["gemm_ms"] / reps, p["gemm_flop"] / p["gemm_ms"] / 1e9, p["orbit_ms"] / reps), flush=True)
