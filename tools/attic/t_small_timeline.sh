#!/bin/bash
# t_small_timeline.sh -- kernel sequence (start, duration) of a plain (T) call of an H2O/cc-pVTZ-shaped system behind an iteration
export TMPDIR=/tmp; cd /tmp && rm -rf /tmp/kt_t
cat > /tmp/t_run.py <<'PY'
import sys, time
sys.path.insert(0, "/root/repo/a-fortran-electronic-structure-program_amd")
from afesp_amd.capi import Engine
eng = Engine(0)
eng.synthetic_init(5, 53, 0.02, 12345, 8)
eng.ccsd_energy()
for it in range(4):
    eng.ccsd_iterate(); eng.ccsd_diis()
    t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(); print("T", (time.perf_counter() - t0) * 1e6)
eng.close()
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_t -- python3 /tmp/t_run.py > /tmp/kt_t.log 2>&1; grep "^T" /tmp/kt_t.log
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/kt_t/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-26:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us +%6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:80]))
PY
