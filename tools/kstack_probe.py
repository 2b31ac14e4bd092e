#!/usr/bin/env python3
"""What stacking two o^3 v^3 ring products along K would buy: the ring product of config 5 with K = 4000 and with K = 8000 (the
summed virtual index twice as long) on the gather kernel.  usage: kstack_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = 20, 200
with Engine(0) as eng:
    for e in (v, 2 * v, 3 * v):
        for (la, da, lb, db, lc, dc) in [("mibe", (o, o, v, e), "mjae", (o, o, v, e), "jbia", (o, v, o, v)),
                                         ("mjae", (o, o, v, e), "iemb", (o, e, o, v), "ijab", (o, o, v, v))]:
            ms = eng.bench_contract(la, da, lb, db, lc, dc, reps=5)
            print("%s,%s>%s  K = %5d: %.3f ms = %.1f TF" % (la, lb, lc, o * e, ms, 2.0 * 4000 * 4000 * o * e / ms / 1e9), flush=True)
