#!/bin/bash
# t_patch_sweep.sh -- plain (T) at config 5 against the LDS-DMA GEMM's patch size (AFESP_TG_PATCH, tiles an XCD's workgroups share;
# default 64) and priority time slice (AFESP_TG_PRIO_SHIFT, default 11): wall time of the 2nd..4th evaluation of a process each
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
run() {
  python3 - "$@" <<PY
import os, sys, time
sys.path.insert(0, os.path.join("$HERE", "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
with Engine(0) as eng:
    eng.synthetic_init(20, 200, 0.005, 12345, 8)
    eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
    eng.do_ccsd_t_spatial_plain()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(); ts.append(time.perf_counter() - t0)
    print("  (T) %.1f %.1f %.1f ms" % tuple(1e3 * t for t in ts), flush=True)
PY
}
for p in 64 16 32 128 256 64; do echo "AFESP_TG_PATCH=$p"; AFESP_TG_PATCH=$p run; done
for s in 9 13 0; do echo "AFESP_TG_PRIO_SHIFT=$s"; AFESP_TG_PRIO_SHIFT=$s run; done
