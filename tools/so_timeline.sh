#!/bin/bash
# so_timeline.sh OUT [OUT_T] -- kernel sequence (start, duration) of the last spin-orbital CCSD iteration at the H2O/cc-pVTZ shape (o=10, v=106) -> OUT;
# with OUT_T: the kernels of the two (T) evaluations behind it (the first builds the operand copies, the second finds them) -> OUT_T
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
OUT_T=""; [ -n "${2:-}" ] && OUT_T="$(realpath -m "$2")"
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt_so
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_so -- python3 "$HERE/tools/so_time.py" > /tmp/kt_so.log 2>&1 || { tail -5 /tmp/kt_so.log; exit 1; }
grep -E "iteration|\(T\)" /tmp/kt_so.log
python3 - "$OUT" "$OUT_T" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kt_so/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last iteration: from the last so_tau_kernel to the kernel before the first (T) launch after it
idx = max(i for i, r in enumerate(rows) if "so_tau_kernel" in r["Kernel_Name"]) - 1
sel = []
for r in rows[idx:]:
    sel.append(r)
    if "lincomb" in r["Kernel_Name"]: break      # the DIIS extrapolation closes the iteration
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
with open(sys.argv[1], "w") as o:
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        o.write("%8.1f us +%7.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    o.write("kernels %d  busy %.1f us  span %.1f us\n" % (len(sel), busy / 1e3, (int(sel[-1]["End_Timestamp"]) - t0) / 1e3))
if len(sys.argv) > 2 and sys.argv[2]:
    # everything behind the iteration: two (T) evaluations (tools/so_time.py)
    rest = rows[idx + len(sel):]
    t0 = int(rest[0]["Start_Timestamp"])
    V, O = 106, 10
    flop = O * (O - 1) * (O - 2) // 6 * 3 * 2.0 * V**3 * (V + O)
    with open(sys.argv[2], "w") as o:
        gemm = []
        for r in rest:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if "tgemm" in r["Kernel_Name"]: gemm.append((e - s) / 1e3)
            o.write("%8.1f us +%7.1f us  grid %8s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", ""), r["Kernel_Name"][:100]))
        if gemm:
            per_eval = sum(gemm) / 2.0
            o.write("tgemm_kernel: %d launches, %.1f us per (T) evaluation; executed flop 120 x 3 x 2 v^3 (v + o) = %.3e -> %.1f TFLOP/s = %.2f of the fp64 MFMA peak (78.6)\n"
                    % (len(gemm), per_eval, flop, flop / per_eval / 1e6, flop / per_eval / 1e6 / 78.6))
PY
