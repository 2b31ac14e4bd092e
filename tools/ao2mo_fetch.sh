#!/bin/bash
# ao2mo_fetch.sh -- bytes fetched past L2 by the GEMM launches of one AO->MO transform (n = 220) for the current environment
# (AFESP_TG_DBG=1: no C stores; AFESP_TG_DYNAMIC=0: static tiles; ...).  usage: ao2mo_fetch.sh LABEL
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pmc_aof
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_aof -- python3 $R/tools/ao2mo_time.py 20 200 1 > /tmp/pmc_aof.log 2>&1 || { tail -3 /tmp/pmc_aof.log; exit 1; }
python3 - "$1" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/pmc_aof/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tgemm_kernel" in r["Kernel_Name"]: agg[(r["Dispatch_Id"])] += float(r["Counter_Value"])
v = [x * 1024 * 2 / 1e9 for x in agg.values()]
print("%s: tgemm launches %d, fetched GB per launch: %s ; per transform %.1f GB" % (sys.argv[1], len(v), " ".join("%.1f" % x for x in v[:7]), sum(v) / 2))
PY
