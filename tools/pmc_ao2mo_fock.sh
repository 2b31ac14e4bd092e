#!/bin/bash
# HBM traffic (rocprofv3 PMC, one counter per pass) of the AO->MO and Fock-build kernels at n = 220.
# usage (on the GPU box): tools/pmc_ao2mo_fock.sh r01   -> gpurun_out/<tag>_pmc_ao2mo_fock.json
set -o pipefail
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pmc_ao_$c -- python3 $R/tools/ao2mo_time.py 20 200 1 > $O/pmc_ao_$c.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pmc_fk_$c -- python3 $R/tools/fock_time.py 220 1 > $O/pmc_fk_$c.log 2>&1 || exit 1
done
python3 - "$O/${TAG}_pmc_ao2mo_fock.json" <<'PY'
import collections, csv, glob, json, sys
out = {"note": "mean per dispatch (and *_per_call: summed over the dispatches of one transform / one Fock build); FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction, DESIGN.md section 5), WRITE_SIZE [KB] x 1024; "
               "durations from the kernel trace of the same pass; n = 220 (npair = 24310)", "kernels": {}}
for tag in ("ao", "fk"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = "/tmp/pmc_%s_%s" % (tag, c)
        dur = collections.defaultdict(list)
        for f in glob.glob(d + "/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                if not any(s in k for s in ("pair_square", "pack_pairs", "gett_kernel", "tgemm_kernel", "tgemm_xform_kernel", "tgemm_mixed_kernel", "fock_")):
                    continue
                e = out["kernels"].setdefault(tag + ": " + k, {"dispatches": len(v)})
                e["fetch_GB" if c == "FETCH_SIZE" else "write_GB"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
                # (the transform's GEMM launches differ in size: also the sum over one call -- the timing scripts make 2 calls)
                e["fetch_GB_per_call" if c == "FETCH_SIZE" else "write_GB_per_call"] = sum(v) / 2 * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
                if k in dur:
                    e["ms_under_pmc"] = sum(dur[k]) / len(dur[k])
                    e["ms_per_call_under_pmc"] = sum(dur[k]) / 2
tot = {"fetch_GB_per_call": 0.0, "write_GB_per_call": 0.0, "ms_per_call_under_pmc": 0.0}
for k, e in out["kernels"].items():
    if k.startswith("ao: "):
        for q in tot: tot[q] += e.get(q, 0.0)
tot["hbm_GB_per_call"] = tot["fetch_GB_per_call"] + tot["write_GB_per_call"]
out["ao2mo_total_per_call"] = tot
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
echo done
