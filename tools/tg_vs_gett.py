#!/usr/bin/env python3
"""E[T]/E(T)/D sums from the LDS-DMA kernel (default) and the grouped gather kernel (AFESP_T_GEMM=gett) against the oracle on
synthetic systems.   usage: tg_vs_gett.py "o,v" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import molecules, orc
from afesp_amd.capi import Engine

for arg in sys.argv[1:]:
    o, v = (int(x) for x in arg.split(","))
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=31 + v)
    cc = orc.OracleCC(o, v, eri, e, 6)
    cc.solve(3, 1e-12, 1e-12)
    ref = np.array(cc.triples(e))
    res = {}
    for mode in ("tg", "gett"):
        os.environ["AFESP_T_GEMM"] = mode
        with Engine(0) as eng:
            eng.ccsd_init(o, v, e, eri, 6)
            eng.do_ccsd_spatial(3, 1e-12, 1e-12)
            res[mode] = np.array(eng.do_ccsd_t_spatial())
    print("o=%d v=%d  tg-ref %.3e  gett-ref %.3e  tg-gett %.3e   ref %s" % (o, v, np.max(np.abs(res["tg"] - ref)), np.max(np.abs(res["gett"] - ref)),
          np.max(np.abs(res["tg"] - res["gett"])), ref[:2]), flush=True)
