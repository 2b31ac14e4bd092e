#!/usr/bin/env python3
"""First-iteration cost of a small system: first context of the process vs a second context in the same process (code objects
and kernel functions already loaded) -- what is per process and what is per context (plans, lanes, scratch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def run(tag):
    t0 = time.perf_counter(); eng = Engine(0); t_ctx = time.perf_counter() - t0
    time.sleep(0.3)                       # let the background preload finish
    t0 = time.perf_counter(); eng.synthetic_init(7, 21, 0.02, 12345, 8); eng.ccsd_energy(); t_init = time.perf_counter() - t0
    its = []
    for _ in range(4):
        t0 = time.perf_counter(); eng.ccsd_iterate(); eng.ccsd_diis(); its.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(); t_t = time.perf_counter() - t0
    # the same shape again in the same context (a scan over geometries): the state is initialised where it lies, its programs stay
    t0 = time.perf_counter(); eng.synthetic_init(7, 21, 0.02, 54321, 8); eng.ccsd_energy(); t_init2 = time.perf_counter() - t0
    its2 = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.ccsd_iterate(); eng.ccsd_diis(); its2.append(time.perf_counter() - t0)
    print(f"{tag}: same shape again: init {t_init2*1e3:.2f} ms, iterations " + " / ".join(f"{x*1e3:.2f}" for x in its2) + " ms")
    eng.close()
    print(f"{tag}: context {t_ctx*1e3:.1f} ms, init {t_init*1e3:.1f} ms, iterations " + " / ".join(f"{x*1e3:.2f}" for x in its) + f" ms, (T) first call {t_t*1e3:.2f} ms")


if __name__ == "__main__":
    run("first context ")
    run("second context")
