#!/usr/bin/env python3
"""The GEMV-like contractions of a config-5 CCSD iteration (one operand of o v^3 = 1.28 GB read once, a small output) under forced
tile shapes and K-slice counts: time and effective HBM rate.  usage: skinny_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = 20, 200
GB = 8.0 * o * v**3 / 1e9
shapes = {
    "r1b  asym.mief x v_vvov.efma -> ia":   ("mief", (o, o, v, v), "efma", (v, v, o, v), "ia", (o, v)),
    "I_vv w_vvov.ebma x t1.me -> ba":        ("ebma", (v, v, o, v), "me", (o, v), "ba", (v, v)),
    "I_ovov t1.je x v_vvov.ebia -> jbia":    ("je", (o, v), "ebia", (v, v, o, v), "jbia", (o, v, o, v)),
    "x_voov v_vvov.beia x t1.je -> bjia":    ("beia", (v, v, o, v), "je", (o, v), "bjia", (v, o, o, v)),
    "r2c  t1.ie x v_vvov.baje -> ijab":      ("ie", (o, v), "baje", (v, v, o, v), "ijab", (o, o, v, v)),
}
eng = Engine(0)
for name, (la, dA, lb, dB, lc, dC) in shapes.items():
    for (tm, tn, sp) in ((0, 0, 0), (1, 1, 0), (1, 4, 0), (4, 1, 0), (2, 2, 0), (1, 1, 64), (1, 1, 256), (4, 1, 64), (4, 1, 256), (1, 4, 256)):
        eng.set_tuning(0, tm, tn, sp)
        try:
            ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=3)
            print("%-40s tm=%d tn=%d split=%3d: %7.3f ms  %5.2f TB/s" % (name, tm, tn, sp, ms, GB / ms), flush=True)
        except Exception as exc:   # noqa: BLE001
            print("%-40s tm=%d tn=%d split=%3d: %s" % (name, tm, tn, sp, str(exc)[:60]), flush=True)
eng.set_tuning(0, 0, 0, 0)
eng.close()
