#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc --output-format csv directory.  usage: pmc_summary.py DIR [name filter]"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if flt in name:
                acc[name[:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in acc.items():
        print(name)
        for c, vals in sorted(cs.items()):
            print(f"   {c:32s} n={len(vals):4d} mean={sum(vals)/len(vals):.6g}")


if __name__ == "__main__":
    main()
