#!/bin/bash
# small_timeline.sh OUT -- kernel timeline (start, duration, queue) of the last CCSD iteration of an H2O/cc-pVTZ-shaped run -> OUT
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt_small
AFESP_GRAPH_AFTER=${AFESP_GRAPH_AFTER:-3} rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_small -- python3 "$HERE/tools/prof_run.py" --o ${PROF_O:-5} --v ${PROF_V:-53} --iters 12 --triples 0 --scale 0.02 ${PROF_EXTRA:-} > /tmp/kt_small.log 2>&1 || { tail -5 /tmp/kt_small.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kt_small/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= end - float(__import__("os").environ.get("WINDOW_NS", "0.55e6"))]
t0 = int(sel[0]["Start_Timestamp"])
with open(sys.argv[1], "w") as o:
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%8.1f us +%6.1f us  q%-3s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
PY
