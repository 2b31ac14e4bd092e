import os, sys, time
sys.path.insert(0, '/root/repo/a-fortran-electronic-structure-program_amd')
from afesp_amd.capi import Engine
eng = Engine(0)
eng.synthetic_init(20, 200, 0.005, 12345, 8)
eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(0, 1); print("first call, 1 triple: %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(0, 1); print("second call, 1 triple: %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(); print("first full call: %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(); print("second full call: %.3f s" % (time.perf_counter() - t0))
