#!/usr/bin/env python3
"""Times contraction shapes of the path under different launcher settings (one process, interleaved)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 20, 200
shapes = {
  "pp_ladder": ("ijef", (o,o,v,v), "efab", (v,v,v,v), "ijab", (o,o,v,v), 2*o*o*v**4),
  "ring":      ("mjae", (o,o,v,v), "iemb", (o,v,o,v), "ijab", (o,o,v,v), 2*o**3*v**3),
  "t_particle":("da", (v,v), "dbc", (v,v,v), "abc", (v,v,v), 2*v**4),
  "ooov":      ("jkef", (o,o,v,v), "efia", (v,v,o,v), "jkia", (o,o,o,v), 2*o**3*v**3),
}
which = sys.argv[1:] or list(shapes)
for name in which:
    la, dA, lb, dB, lc, dC, fl = shapes[name]
    for (gm, tm, tn) in [(0,4,4), (0,16,8), (0,8,16)]:
        eng.set_tuning(gm, tm, tn, 0)
        ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=3)
        print(f"{name:12s} gm={gm:2d} tm={tm} tn={tn}: {ms:9.3f} ms  {fl/ms/1e9:7.2f} TF", flush=True)
eng.close()
