#!/usr/bin/env python3
"""(T) variants at config-5 extents (or `o v`): wall time of the plain / full (D sums) / completely renormalised evaluations and
the engine's own HIP-event split into grouped-GEMM and orbit-kernel time (afesp_profile).
usage: t_time.py [o v [reps]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def main():
    o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.005, 12345, 8)
        eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
        for name, fn in (("plain", eng.do_ccsd_t_spatial_plain), ("full", eng.do_ccsd_t_spatial)):
            fn()
            eng.profile(True)
            t0 = time.perf_counter()
            for _ in range(reps): out = fn()
            dt = (time.perf_counter() - t0) / reps
            p = eng.profile(False)
            print("%-6s %.2f ms  profile %s" % (name, dt * 1e3, p))
            print("       ", ["%.12f" % x for x in out])
        eng.build_cr_intermediates()
        eng.do_ccsd_t_spatial_cr()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps): out = eng.do_ccsd_t_spatial_cr()
        dt = (time.perf_counter() - t0) / reps
        print("%-6s %.2f ms  profile %s" % ("cr", dt * 1e3, eng.profile(False)))
        print("       ", ["%.12f" % x for x in out])


if __name__ == "__main__":
    main()
