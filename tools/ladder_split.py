#!/usr/bin/env python3
"""pp-ladder time at config 5 against the number of K slices of its two pair products (AFESP_PP_SPLIT; 0 = the launcher's
wave-quantisation score).  Round 2: 0 -> 7.05 ms (the launcher picks 3), 2: 11.0, 3: 7.05, 4: 8.0, 6: 7.17, 8: 7.25."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
with Engine(0) as eng:
    eng.synthetic_init(20, 200, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate()
    for sp in ("0", "2", "3", "4", "5", "6", "8", "10", "0"):
        os.environ["AFESP_PP_SPLIT"] = sp
        print("split", sp, "ladder %.3f ms" % eng.time_pp_ladder(20))
