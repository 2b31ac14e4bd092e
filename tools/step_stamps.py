import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo/a-fortran-electronic-structure-program_amd')
from afesp_amd.capi import Engine
eng = Engine(0)
M, N, K = 4096, 128 * 16 * 8, 224     # 8 rounds of 256 tiles, 14 steps each
eng.set_tuning(0, 0, 0, 0)
eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=2)
eng.set_tuning(4 << 17, 0, 0, 0)
eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=1)
eng.set_tuning(0, 0, 0, 0)
G = 8 * 14
buf = (C.c_longlong * (4 * G))()
eng.L.afesp_debug_read_ws.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int64]
eng.L.afesp_debug_read_ws(eng.h, buf, 4 * G)
st = np.array(buf[:]).reshape(G, 4)
start = st[:, 0]
d = np.diff(start) / 100.0          # s_memtime ticks at 100 MHz -> us
for tile in range(8):
    row = d[tile * 14:(tile + 1) * 14]
    last = st[tile * 14 + 13]
    nxt = st[tile * 14 + 14, 0] if tile < 7 else last[3]
    print("tile %d steps(100 cyc): %s | step13 body %.0f  stores issued %.0f  drained(waitcnt 0) %.0f  to next step %.0f" % (
        tile, " ".join("%.0f" % x for x in row[:13]), (last[1] - last[0]) / 100.0, (last[2] - last[1]) / 100.0, (last[3] - last[2]) / 100.0, (nxt - last[3]) / 100.0))
