#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per (T) GEMM launch from two rocprofv3 --pmc passes over bench.py; writes the
entry bench.py reports as roofline.traffic.  usage: pmc_bench_traffic.py <workload> <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys
wl, fdir, wdir, outp = sys.argv[1:5]
def avg(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if ("tgemm_kernel" in r["Kernel_Name"] or ("gett_kernel" in r["Kernel_Name"] and ", true, false>" in r["Kernel_Name"])) and r["Counter_Name"] == counter]   # <..., GRP = true, RAG = false>: the grouped (T) launches
    return sum(vals) / len(vals), len(vals)
fetch, nf = avg(fdir, "FETCH_SIZE")
write, nw = avg(wdir, "WRITE_SIZE")
try:
    out = json.load(open(outp))
except Exception:
    out = {}
out[wl + "_t_gemm"] = {"kernel": "tgemm_kernel (or gett_kernel<..., GRP = true> under AFESP_T_GEMM=gett): the grouped (T) GEMM launches, one per chunk; the run is bench.py --steps-only, so the AO->MO leg's launches of the same kernel are not in it", "dispatches": nf,
                       "fetch_bytes_per_launch": fetch * 1024 * 2, "write_bytes_per_launch": write * 1024,
                       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
                       "correction": "FETCH_SIZE [KB] x1024 x2 (gfx950 counts 128-B requests as 64 B; verified on an 8 B/lane stream of known size, r01_pmc_*), WRITE_SIZE [KB] x1024",
                       "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --workload {wl} --steps 1 --warmup 0 --no-cpu-baseline --steps-only (two passes)"}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out[wl + "_t_gemm"], indent=1))
