#!/bin/bash
# FETCH_SIZE (bytes past L2) and duration per (T) launch of tgemm_kernel at config 5 against the tile walk's knobs: tiles per XCD patch
# (AFESP_TG_PATCH), tickets (AFESP_TG_DYNAMIC), occupied block size of the enumeration (AFESP_T_BLOCK: columns per group = block x v).
# One rocprofv3 --pmc FETCH_SIZE pass per setting.  usage (GPU box): tools/tg_traffic_sweep.sh OUT
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/tg_traffic_sweep.txt}
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
run() {   # label, env...
  local label="$1"; shift
  rm -rf /tmp/pmc_sw
  env "$@" true 2>/dev/null
  ( export "$@"; rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_sw -- python3 $R/bench.py --workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-live-pmc --steps-only > /tmp/pmc_sw.log 2>&1 )
  python3 - "$label" >> "$OUT" <<'PY'
import csv, glob, sys
label = sys.argv[1]
f = glob.glob("/tmp/pmc_sw/*/*counter_collection.csv")
k = glob.glob("/tmp/pmc_sw/*/*kernel_trace.csv")
if not f or not k:
    print(label, "no data"); sys.exit(0)
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "tgemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(k[0])) if "tgemm_kernel" in r["Kernel_Name"]]
print("%-40s launches %3d  fetch %6.2f GB per launch  %7.3f ms per launch (under PMC)" % (label, len(vals), sum(vals) / len(vals) * 2048 / 1e9, sum(dur) / len(dur)))
PY
}
run "default (patch 64, tickets, block 5)" AFESP_DUMMY=1
for p in 16 32 128 256; do run "patch $p" AFESP_TG_PATCH=$p; done
run "static dealing (no tickets)" AFESP_TG_DYNAMIC=0
run "patch 32, no tickets" AFESP_TG_PATCH=32 AFESP_TG_DYNAMIC=0
for b in 4 6 7; do run "block $b" AFESP_T_BLOCK=$b; done
run "prio off" AFESP_TG_PRIO_SHIFT=0
cat "$OUT"
