#!/usr/bin/env python3
"""Stage times of one post-SCF calculation through the C-ABI on synthetic AO integrals (default o=20, v=200): context, AO->MO + MP2,
CCSD initialisation, iterations, (T) -- first calls included, as a user of the library sees them."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
nit = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = o + v
q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
T = {}
def stage(name, t0): T[name] = time.perf_counter() - t0
t0 = time.perf_counter(); eng = Engine(0); stage("context", t0)
t0 = time.perf_counter(); eng.synthetic_ao(n, 0.0005, 777); stage("AO integrals generated on the device", t0)
rng = np.random.default_rng(3); d = rng.standard_normal((n, n)); d = d + d.T
t0 = time.perf_counter(); eng.build_fock(n, d, d); stage("first Fock build (squares the integrals up)", t0)
t0 = time.perf_counter(); eng.build_fock(n, d, d); stage("second Fock build", t0)
t0 = time.perf_counter(); emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False); stage("AO->MO + MP2", t0)
t0 = time.perf_counter(); eng.ccsd_init(o, v, e, None, 8); eng.ccsd_energy(); stage("ccsd_init + MP1 energy", t0)
per = []
for it in range(nit):
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); eng.ccsd_diis(); per.append(time.perf_counter() - t0)
T["CCSD iterations: first / second / median of the rest"] = (per[0], per[1], float(np.median(per[2:])))
t0 = time.perf_counter(); et = eng.do_ccsd_t_spatial_plain(); stage("(T), first call", t0)
t0 = time.perf_counter(); et = eng.do_ccsd_t_spatial_plain(); stage("(T), second call", t0)
eng.close()
for k, x in T.items():
    print("%-60s %s" % (k, " / ".join("%.1f ms" % (y * 1e3) for y in x) if isinstance(x, tuple) else "%.1f ms" % (x * 1e3)))
print("E(MP2) %.10f  last CCSD energy %.10f  E[T] %.10f" % (emp2, r[0], et[0]))
