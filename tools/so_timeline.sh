#!/bin/bash
# so_timeline.sh OUT -- kernel sequence (start, duration) of the last spin-orbital CCSD iteration at the H2O/cc-pVTZ shape (o=10, v=106) -> OUT
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt_so
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_so -- python3 "$HERE/tools/so_time.py" > /tmp/kt_so.log 2>&1 || { tail -5 /tmp/kt_so.log; exit 1; }
grep -E "iteration|\(T\)" /tmp/kt_so.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kt_so/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last iteration: from the last so_tau_kernel to the kernel before the first (T) launch after it
idx = max(i for i, r in enumerate(rows) if "so_tau_kernel" in r["Kernel_Name"]) - 1
sel = []
for r in rows[idx:]:
    if "triples" in r["Kernel_Name"] or "tgemm" in r["Kernel_Name"]: break
    sel.append(r)
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
with open(sys.argv[1], "w") as o:
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        o.write("%8.1f us +%7.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    o.write("kernels %d  busy %.1f us  span %.1f us\n" % (len(sel), busy / 1e3, (int(sel[-1]["End_Timestamp"]) - t0) / 1e3))
PY
