import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo/a-fortran-electronic-structure-program_amd")
from afesp_amd.capi import Engine
n, o = 220, 20
q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(n - o) / (n - o - 1)])
eng = Engine(0)
if len(sys.argv) > 1: time.sleep(float(sys.argv[1]))
eng.synthetic_ao(n, 0.02, 777)
for r in range(3):
    t0 = time.perf_counter()
    emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)
    print("call", r, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
eng.close()
