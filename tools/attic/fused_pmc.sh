#!/bin/bash
# fused_pmc.sh OUT -- PMC counters of the product launches of the launch-fused iteration (o=5, v=53), one rocprofv3 pass per counter set
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/fpmc_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/fpmc_$i -- python3 "$HERE/tools/prof_run.py" --o 5 --v 53 --iters 8 --triples 0 --scale 0.02 > /tmp/fpmc_$i.log 2>&1 || { tail -5 /tmp/fpmc_$i.log; continue; }
  python3 - /tmp/fpmc_$i >> "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
g = collections.OrderedDict()
for r in rows:
    if "fused_gemm" not in r["Kernel_Name"]: continue
    g.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(g)[-9:]           # the last three iterations: stages 1, 3, 5 in turn
for st in range(3):
    sel = [g[d] for k, d in enumerate(ids) if k % 3 == st]
    names = sorted(sel[0])
    print("stage %d: " % (2 * st + 1) + "  ".join("%s %.4g" % (n, sum(s[n] for s in sel) / len(sel)) for n in names))
PY
done
cat "$OUT"
