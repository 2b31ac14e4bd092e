#!/usr/bin/env python3
"""Every tensor product of one CCSD iteration alone on the device (AFESP_CONTRACT_TRACE=1): labels, GEMM extents, time, rate.
usage: contract_trace.py [--o 20 --v 200]   (lines on stderr, the last iteration is the warm one)"""
import argparse, os, sys
os.environ["AFESP_CONTRACT_TRACE"] = "1"
os.environ["AFESP_NO_GRAPH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
ap = argparse.ArgumentParser()
ap.add_argument("--o", type=int, default=20); ap.add_argument("--v", type=int, default=200)
ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
eng = Engine(0)
eng.synthetic_init(a.o, a.v, 0.005, 12345, 8)
for it in range(a.iters):
    sys.stderr.write("==== iteration %d\n" % it); sys.stderr.flush()
    eng.ccsd_iterate(); eng.ccsd_diis()
eng.close()
