#!/usr/bin/env python3
"""Kernel sequence of the last CCSD iteration in a rocprofv3 kernel trace of tools/prof_run.py: start offset, duration,
grid, kernel.  usage: iter_timeline.py <rocprof output dir> [window_ms]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 33.0
end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= end - win * 1e6]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us %9.1f us  grid %9s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")),
                                            r["Kernel_Name"][:100]))
