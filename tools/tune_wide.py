import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 20, 200
shapes = {
  "pp_ladder": ("ijef", (o,o,v,v), "efab", (v,v,v,v), "ijab", (o,o,v,v), 2*o*o*v**4),
  "ring":      ("mjae", (o,o,v,v), "iemb", (o,v,o,v), "ijab", (o,o,v,v), 2*o**3*v**3),
  "t_fused":   ("kbc", (v+o,v,v), "kap", (v+o,v,48), "abcp", (v,v,v,48), 2*(v+o)*v**3*48),
  "ooov":      ("jkef", (o,o,v,v), "efia", (v,v,o,v), "jkia", (o,o,o,v), 2*o**3*v**3),
}
for name, (la, dA, lb, dB, lc, dC, fl) in shapes.items():
    for wide in (0, 1):
        for (tm, tn) in [(4,4), (4,2), (16,8)]:
            eng.set_tuning(0 if wide else 0x10000, tm, tn, 0)
            ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=3)
            print(f"{name:10s} wide={wide} tm={tm:2d} tn={tn:2d}: {ms:9.3f} ms  {fl/ms/1e9:7.2f} TF", flush=True)
eng.close()
