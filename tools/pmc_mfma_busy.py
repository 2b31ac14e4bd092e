#!/usr/bin/env python3
"""MFMA-pipe busy fraction of the grouped (T) GEMM launches from the summarised PMC passes:
SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs) divided by GRBM_GUI_ACTIVE / 8 (the counter sums its eight XCD instances).
Adds `mfma_busy_frac` to <dir>/<tag>_traffic.json, which bench.py reports as roofline.mfma_busy.
usage: pmc_mfma_busy.py <tag> <dir with the <tag>_pmc_*.json summaries>"""
import json, os, sys
tag, d = sys.argv[1], sys.argv[2]
tp = os.path.join(d, tag + "_traffic.json")
out = json.load(open(tp))
for wl in ("cfg5", "h2o_tz"):
    try:
        busy = json.load(open(os.path.join(d, f"{tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_{wl}.json")))["counters"]
        act = json.load(open(os.path.join(d, f"{tag}_pmc_GRBM_GUI_ACTIVE_{wl}.json")))["counters"]
    except (OSError, KeyError):
        continue
    grp = lambda rows: [r for r in rows if ("tgemm_kernel" in r["name"] or ("gett_kernel" in r["name"] and ", true, false>" in r["name"]))]   # GRP = true, RAG = false
    b, a = grp(busy), grp(act)
    if not b or not a or (wl + "_t_gemm") not in out:
        continue
    nb = sum(r["dispatches"] for r in b)
    cyc = sum(r["mean_per_dispatch"]["SQ_VALU_MFMA_BUSY_CYCLES"] * r["dispatches"] for r in b) / nb
    gui = sum(r["mean_per_dispatch"]["GRBM_GUI_ACTIVE"] * r["dispatches"] for r in a) / sum(r["dispatches"] for r in a)
    out[wl + "_t_gemm"]["mfma_busy_frac"] = cyc / (256 * 4) / (gui / 8)
    out[wl + "_t_gemm"]["mfma_busy_source"] = "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, separate rocprofv3 --pmc passes of the same command"
json.dump(out, open(tp, "w"), indent=1)
print({k: v.get("mfma_busy_frac") for k, v in out.items()})
