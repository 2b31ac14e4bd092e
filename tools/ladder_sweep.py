#!/usr/bin/env python3
"""pp-ladder time at config 5 against forced tile codes / K slices (afesp_set_tuning; 0 = the launcher's own choice).  usage: ladder_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
with Engine(0) as eng:
    eng.synthetic_init(20, 200, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate()
    for (tm, tn, sp) in [(0, 0, 0), (4, 2, 0), (4, 2, 2), (4, 2, 3), (4, 2, 4), (4, 4, 0), (4, 4, 2), (4, 4, 3), (16, 8, 3), (16, 8, 2), (8, 8, 3), (8, 16, 3)]:
        eng.set_tuning(0, tm, tn, sp)
        try:
            print("tm %2d tn %2d split %d: pp-ladder %.3f ms" % (tm, tn, sp, eng.time_pp_ladder(5)), flush=True)
        except Exception as e:
            print("tm %d tn %d split %d: %s" % (tm, tn, sp, e), flush=True)
    eng.set_tuning(0, 0, 0, 0)
