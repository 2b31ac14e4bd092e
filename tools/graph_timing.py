import sys, os, time
sys.path.insert(0, '/root/repo/a-fortran-electronic-structure-program_amd')
from afesp_amd.capi import Engine
eng = Engine(0)
eng.synthetic_init(5, 53, 0.02, 12345, 8)
eng.ccsd_energy()
for it in range(8):
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); t1 = time.perf_counter(); eng.ccsd_diis(); t2 = time.perf_counter()
    print(it, "iterate %.3f ms  diis %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), r[0])
