#!/usr/bin/env python3
"""Larger than config 5 (default o=30, v=300: v^4 = 65 GB): two CCSD iterations, then (T) on a few shards, checking the
size-independent properties (shard additivity, finiteness, t2 symmetry) -- a guard against 32-bit overflows in the plans."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 300)
eng = Engine(0)
t0 = time.perf_counter(); eng.synthetic_init(o, v, 0.002, 12345, 4); print("init %.1f s" % (time.perf_counter() - t0), flush=True)
print("MP1", eng.ccsd_energy())
for it in range(2):
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); eng.ccsd_diis(); print("iter", it, r, "%.3f s" % (time.perf_counter() - t0), flush=True)
nt = eng.ntriples()
t0 = time.perf_counter(); full = eng.do_ccsd_t_spatial(); dt = time.perf_counter() - t0
fl = nt * 12.0 * v**3 * (v + o)
print("(T) %.3f s  %.1f TFLOP/s (symmetric count)" % (dt, fl / dt / 1e12), full, flush=True)
cuts = [0, nt // 3, nt // 2 + 7, nt]
parts = sum(eng.do_ccsd_t_spatial(a, b) for a, b in zip(cuts[:-1], cuts[1:]))
print("shards", parts, "max rel diff %.2e" % np.max(np.abs(parts - full) / np.abs(full)))
assert np.all(np.isfinite(full)) and np.max(np.abs(parts - full) / np.abs(full)) < 1e-10
eng.close()
print("ok")
