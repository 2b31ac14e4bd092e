#!/usr/bin/env python3
"""One contraction over forced tile codes / K slices of the gather kernel (afesp_set_tuning): time per launch.
usage: tile_sweep.py o v [labels "la,lb>lc" with index letters from ijklmn (occupied) and abcdef (virtual)]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine

o, v = int(sys.argv[1]), int(sys.argv[2])
spec = sys.argv[3] if len(sys.argv) > 3 else "imbe,mjea>bjia"
la, rest = spec.split(",")
lb, lc = rest.split(">")
dim = lambda l: [o if ch in "ijklmn" else v for ch in l]
with Engine(0) as eng:
    for tm, tn, split in [(0, 0, 0), (4, 4, 1), (4, 4, 2), (4, 4, 4), (4, 4, 8), (4, 2, 1), (2, 4, 1), (4, 2, 2), (2, 2, 1), (2, 2, 2), (2, 1, 1), (1, 2, 1),
                          (1, 1, 1), (8, 8, 1), (8, 8, 2), (8, 8, 4)]:
        eng.set_tuning(0, tm, tn, split)
        ms = eng.bench_contract(la, dim(la), lb, dim(lb), lc, dim(lc), 20)
        fl = 2.0
        for ch in set(la + lb + lc): fl *= (o if ch in "ijklmn" else v)
        print(f"{spec} o={o} v={v} tm={tm} tn={tn} split={split}: {ms*1e3:8.1f} us  {fl/ms*1e-9:6.2f} TF", flush=True)
    eng.set_tuning(0, 0, 0, 0)
