import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 5, 53
shapes = {
  "pp_ladder": ("ijef", (o,o,v,v), "efab", (v,v,v,v), "ijab", (o,o,v,v)),
  "ring":      ("mjae", (o,o,v,v), "iemb", (o,v,o,v), "ijab", (o,o,v,v)),
  "t1_vovv":   ("ie", (o,v), "ejab", (v,o,v,v), "ijab", (o,o,v,v)),
  "Ivv2":      ("mneb", (o,o,v,v), "mnea", (o,o,v,v), "ba", (v,v)),
  "Ivo":       ("miea", (o,o,v,v), "me", (o,v), "ai", (v,o)),
  "ooov":      ("jkef", (o,o,v,v), "efia", (v,v,o,v), "jkia", (o,o,o,v)),
  "t2Ivv":     ("ijae", (o,o,v,v), "eb", (v,v), "ijab", (o,o,v,v)),
}
for name, (la, dA, lb, dB, lc, dC) in shapes.items():
    for split in (0, 1):
        eng.set_tuning(0, 0, 0, split)
        ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=50)
        print(f"{name:10s} split={'auto' if split == 0 else 'off '}: {ms*1e3:8.2f} us", flush=True)
eng.close()
