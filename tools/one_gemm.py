import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
shape = sys.argv[1] if len(sys.argv) > 1 else "ring"
tm, tn = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 0)
o, v = 20, 200
shapes = {
  "pp_ladder": ("ijef", (o,o,v,v), "efab", (v,v,v,v), "ijab", (o,o,v,v), 2*o*o*v**4),
  "ring":      ("mjae", (o,o,v,v), "iemb", (o,v,o,v), "ijab", (o,o,v,v), 2*o**3*v**3),
}
la, dA, lb, dB, lc, dC, fl = shapes[shape]
eng = Engine(0)
eng.set_tuning(0, tm, tn, 0)
ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=3)
print(f"{shape} tm={tm} tn={tn}: {ms:.3f} ms {fl/ms/1e9:.2f} TF")
eng.close()
