import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
o, v = 20, 200
for (la, da, lb, db, lc, dc) in [("mibe", (o, o, v, v), "mjae", (o, o, v, v), "jbia", (o, v, o, v)), ("mjae", (o, o, v, v), "iemb", (o, v, o, v), "ijab", (o, o, v, v))]:
    ms = eng.bench_contract(la, da, lb, db, lc, dc, reps=3)
    buf = (C.c_ulonglong * (256 * 8 * 4))()
    eng.L.afesp_debug_stamps(buf, 256 * 8 * 4)
    a = np.array(buf, dtype=np.float64).reshape(256, 8, 4)
    a = a[a[:, :, 3].min(axis=1) > 0]
    raw = a[..., 3].astype(np.uint64)
    a[..., 3] = (raw & np.uint64((1 << 20) - 1)).astype(np.float64)
    tot, bar, st, n = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    print(f"{la},{lb}>{lc}: {ms*1e3:.0f} us = {2.0*4000**3/ms/1e9:.1f} TF  workgroups {a.shape[0]}  steps/wave {n.mean():.0f}  cycles/step {np.mean(tot/n):.0f} (floor 8192)  barrier {np.mean(bar/n):.0f}  stash {np.mean(st/n):.0f}; implied clock {np.mean(tot)/ (ms*1e-3)/1e9:.2f} GHz if the waves ran the whole time")
eng.close()
