#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: GPU-busy time and idle gaps between kernels over the timed steps."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / 1e3
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print(f"kernels {len(tail)}  busy {busy:.1f} us  span {span:.1f} us  idle {span - busy:.1f} us ({100 * (span - busy) / span:.0f} %)")

import collections
by = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    k = r["Kernel_Name"][:100]
    by[k][0] += 1
    by[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, tt) in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{c:5d} x {tt / c:8.2f} us = {tt:9.1f} us  {k}")
