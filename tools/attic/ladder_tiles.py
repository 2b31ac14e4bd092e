#!/usr/bin/env python3
"""Tile shapes of the two pair products of the pp-ladder (AFESP_PP_TILES="tm,tn,split,tm,tn,split", csrc/ccsd.hip): time per ladder
call and the residual with its ladder part for each setting, compared bit by bit with the launcher's own choice.
usage: ladder_tiles.py [o v] [setting ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def main():
    args = sys.argv[1:]
    o, v = 20, 200
    if len(args) >= 2 and args[0].isdigit():
        o, v, args = int(args[0]), int(args[1]), args[2:]
    settings = args or ["0,0,0,0,0,0", "16,7,0,0,0,0", "0,0,0,16,6,0", "16,7,0,16,6,0", "16,7,0,16,7,0", "16,8,0,16,8,0"]
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.005, 12345, 8)
        eng.ccsd_energy(); eng.ccsd_iterate()
        ref = None
        for s in settings:
            os.environ["AFESP_PP_TILES"] = s
            ms = eng.time_pp_ladder(5)
            r2 = eng.tensor("r2")
            if ref is None:
                ref = r2
            print(f"AFESP_PP_TILES={s:18s} {ms:8.3f} ms per ladder   max |r2 - r2(first setting)| = {np.max(np.abs(r2 - ref)):.3e}", flush=True)


if __name__ == "__main__":
    main()
