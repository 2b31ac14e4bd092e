import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 20, 200
la, dA, lb, dB, lc, dC, fl = ("mjae", (o,o,v,v), "iemb", (o,v,o,v), "ijab", (o,o,v,v), 2*o**3*v**3)
for ab in [0, 1, 2, 4, 7, 0]:
    eng.set_tuning(ab << 16, 4, 4, 0)
    ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=5)
    print(f"ring 8-wave ablate={ab} (1=no global loads 2=no LDS stores 4=no barrier): {ms:8.3f} ms {fl/ms/1e9:7.2f} TF-equivalent", flush=True)
eng.close()
