#!/usr/bin/env python3
"""Small driver for rocprofv3: N CCSD iterations and a slab of (T) triples on a synthetic system."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
ap = argparse.ArgumentParser()
ap.add_argument("--o", type=int, default=20); ap.add_argument("--v", type=int, default=200)
ap.add_argument("--iters", type=int, default=2); ap.add_argument("--triples", type=int, default=32)
ap.add_argument("--no-diis", action="store_true"); ap.add_argument("--ladder", type=int, default=0); ap.add_argument("--scale", type=float, default=0.005)
a = ap.parse_args()
eng = Engine(0)
eng.synthetic_init(a.o, a.v, a.scale, 12345, 8)
print("MP1", eng.ccsd_energy())
for it in range(a.iters):
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); (None if a.no_diis else eng.ccsd_diis()); print("iter", it, r, time.perf_counter() - t0)
if a.ladder:
    print("pp-ladder ms", eng.time_pp_ladder(a.ladder))
if a.triples:
    t0 = time.perf_counter(); out = eng.do_ccsd_t_spatial(0, a.triples); print("(T) slab", out, time.perf_counter() - t0)
eng.close()
