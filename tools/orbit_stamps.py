#!/usr/bin/env python3
"""Where a workgroup of the (T) orbit kernel spends its cycles (diagnostic build: triples.hip compiled with -DAFESP_ORBIT_STAMPS,
e.g. `EXTRA=-DAFESP_ORBIT_STAMPS tools/build_orbit_variant.sh stamps`).
usage: AFESP_LIBRARY=build/ab/libafesp_stamps.so orbit_stamps.py [o v]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine

NAMES = ["prologue (meta, patches) up to barrier", "term 0: barrier+park+barrier (HBM wait)", "term 0: permuted reads",
         "term 1: barrier+park+barrier", "term 1: reads", "term 2: barrier+park+barrier", "term 2: reads", "energy phase",
         "reduction + store"]


def main():
    o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.005, 12345, 8)
        eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
        buf = (C.c_ulonglong * 16)()
        for name, fn in (("plain", eng.do_ccsd_t_spatial_plain), ("full", eng.do_ccsd_t_spatial)):
            fn()
            eng.L.afesp_debug_stamps(buf, -10)
            eng.profile(True); fn(); p = eng.profile(False)
            eng.L.afesp_debug_stamps(buf, -10)
            a = np.array(buf[:10], dtype=np.float64)
            n = a[9]
            print("%s: orbit %.2f ms in %d launches, %d workgroups, %.0f cycles per workgroup" % (name, p["orbit_ms"], p["orbit_launches"], n, a[:9].sum() / max(n, 1)))
            for k in range(9):
                print("   %-45s %8.0f cycles  %5.1f %%" % (NAMES[k], a[k] / max(n, 1), 100 * a[k] / max(a[:9].sum(), 1)))


if __name__ == "__main__":
    main()
