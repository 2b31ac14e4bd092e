#!/usr/bin/env python3
"""Plain (T) at mid sizes: wall time of the second call and the rate on the symmetric flop count [o(o+1)(o+2)/6] 12 v^3 (v+o)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(8, 80), (10, 100), (12, 120), (14, 140), (16, 160), (18, 180), (20, 200)]
for o, v in shapes:
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.005, 12345, 8)
        eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
        eng.do_ccsd_t_spatial_plain()
        t0 = time.perf_counter(); out = eng.do_ccsd_t_spatial_plain(); dt = time.perf_counter() - t0
        fl = o * (o + 1) * (o + 2) / 6 * 12.0 * v**3 * (v + o)
        print(f"o={o} v={v}: (T) {dt*1e3:9.3f} ms  {fl/dt/1e12:6.2f} TFLOP/s (symmetric count) = {fl/dt/78.6e12:5.3f} of peak", flush=True)
