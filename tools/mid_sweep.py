#!/usr/bin/env python3
"""Tile / K-slice sweep (afesp_set_tuning) of the mid-size products of a config-5 CCSD iteration: what the launcher picks by
itself (first column) against forced tile codes and slice counts.  usage: mid_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = 20, 200
D = {"o": o, "v": v}
def dims(lab, kinds): return [D[k] for k in kinds]
cases = [  # labels, index kinds
    ("klef", "oovv", "ijef", "oovv", "klij", "oooo"),
    ("ijmn", "oooo", "mnab", "oovv", "ijab", "oovv"),
    ("mneb", "oovv", "mnea", "oovv", "ba", "vv"),
    ("ijae", "oovv", "eb", "vv", "ijab", "oovv"),
    ("ie", "ov", "baje", "vvov", "ijab", "oovv"),
    ("ebma", "vvov", "me", "ov", "ba", "vv"),
    ("mibj", "oovo", "ma", "ov", "jbia", "ovov"),
]
with Engine(0) as eng:
    for la, ka, lb, kb, lc, kc in cases:
        row = []
        for (tm, tn, sp) in [(0, 0, 0), (4, 4, 0), (8, 8, 0), (16, 8, 0), (2, 4, 0), (4, 2, 0), (4, 1, 0), (0, 0, 2), (0, 0, 4), (0, 0, 8), (0, 0, 16), (0, 0, 32), (0, 0, 64)]:
            eng.set_tuning(0, tm, tn, sp)
            try:
                ms = eng.bench_contract(la, dims(la, ka), lb, dims(lb, kb), lc, dims(lc, kc), 5)
                row.append("%d/%d/%d:%.0f" % (tm, tn, sp, ms * 1e3))
            except Exception as e:
                row.append("%d/%d/%d:err" % (tm, tn, sp))
        eng.set_tuning(0, 0, 0, 0)
        print("%s,%s>%s  " % (la, lb, lc) + "  ".join(row), flush=True)
