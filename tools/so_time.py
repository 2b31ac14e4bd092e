import sys, os, time
sys.path.insert(0, '/root/repo/a-fortran-electronic-structure-program_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, orc, molecules
from afesp_amd.capi import Engine
si, ints, res, _ = molecules.load("h2o-cc-pvdz")
n, nel = ints.nbasis, ints.nel
eng = Engine(0)
t=time.time(); e_mp2, eri_mo = eng.do_mp2_spatial(n, nel // 2, res.canon_coeff, res.canon_levels, ints.eri); print("mp2 %.2f"%(time.time()-t))
t=time.time(); eng.init_cc_spinorb(n, nel, res.canon_levels, None, 8); print("init %.2f"%(time.time()-t))
t=time.time(); nit, en, rm = eng.do_ccsd_spinorb(60, 1e-9, 1e-9); print("solve %d its %.2f"%(nit, time.time()-t))
t=time.time(); eng.init_cc_spinorb(n, nel, res.canon_levels, None, 8); nit, en, rm = eng.do_ccsd_spinorb(60, 1e-9, 1e-9); print("solve again %d its %.2f"%(nit, time.time()-t))
t=time.time(); et = eng.do_ccsd_t_spinorb(); print("(T) %.2f"%(time.time()-t), et)
t=time.time(); et = eng.do_ccsd_t_spinorb(); print("(T) again %.2f"%(time.time()-t), et)
t=time.time(); so = orc.OracleSO(n, nel, orc.ao2mo(n, res.canon_coeff, ints.eri), res.canon_levels, 8); print("oracle create %.2f"%(time.time()-t))
t=time.time(); onit, oen, orm = so.solve(60, 1e-9, 1e-9); print("oracle solve %.2f"%(time.time()-t))
t=time.time(); so.triples(); print("oracle T %.2f"%(time.time()-t))
