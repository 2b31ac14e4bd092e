#!/usr/bin/env python3
"""pp-ladder time against the tile-walk group size of the gather kernel (afesp_set_tuning group_m): with two column tiles per row
panel the group decides whether both tiles of a panel of V+- run on one XCD (one L2) at the same time.  usage: ladder_gm.py [o v] [gm ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
args = sys.argv[1:]
o, v = 20, 200
if len(args) >= 2:
    o, v, args = int(args[0]), int(args[1]), args[2:]
gms = [int(x) for x in args] or [0, 1, 2, 3, 4, 5, 6, 8, 10, 11, 16, 32, 0]
with Engine(0) as eng:
    eng.synthetic_init(o, v, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate()
    ref = None
    for gm in gms:
        eng.set_tuning(group_m=gm)
        ms = eng.time_pp_ladder(5)
        r2 = eng.tensor("r2")
        ref = r2 if ref is None else ref
        print(f"group_m={gm:3d}  {ms:8.3f} ms per ladder   max |r2 - r2(first)| = {np.max(np.abs(r2 - ref)):.3e}", flush=True)
    eng.set_tuning()
