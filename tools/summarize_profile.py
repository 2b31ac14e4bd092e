#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace and/or counter collection) into the small tables kept under profiles/."""
import collections, csv, glob, json, sys
d = sys.argv[1]
out = {}
for f in glob.glob(d + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [{"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                            "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:25]]
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out["by_kernel_and_grid"] = sorted(({"name": k[0][:110], "grid": k[1], "calls": len(v), "avg_us": sum(v) / len(v),
                                         "total_ms": sum(v) / 1e3} for k, v in by.items()), key=lambda x: -x["total_ms"])[:25]
for f in glob.glob(d + "/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[(r["Kernel_Name"][:110], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.setdefault("counters", []).extend({"name": k[0], "grid": k[1], "dispatches": max(len(x) for x in c.values()),
                                           "mean_per_dispatch": {n: sum(x) / len(x) for n, x in c.items()}} for k, c in agg.items())
json.dump(out, sys.stdout, indent=1)
