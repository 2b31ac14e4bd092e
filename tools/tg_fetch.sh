#!/bin/bash
# tg_fetch.sh BIN [args...] -- FETCH_SIZE (L2 misses, x2 per the gfx950 correction) and WRITE_SIZE per tgemm_kernel launch of a stand-alone
# check binary (tools/tgemm_check.hip), GB.  usage on the GPU box: tools/tg_fetch.sh tools/tgemm_check_bin big 40000 224 1000 12
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
BIN="$HERE/$1"; shift
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tgf_$c
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/tgf_$c -- "$BIN" "$@" > /tmp/tgf_$c.log 2>&1 || { echo "rocprofv3 failed"; tail -3 /tmp/tgf_$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob
out = {}
for c, f in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    vals = []
    for p in glob.glob("/tmp/tgf_%s/*/*counter_collection.csv" % c):
        vals += [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if "tgemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = sum(vals) / max(len(vals), 1) * f / 1e9
print("per launch: fetched %.2f GB, written %.2f GB" % (out["FETCH_SIZE"], out["WRITE_SIZE"]))
PY
