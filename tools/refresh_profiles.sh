#!/bin/bash
# Regenerates everything under profiles/ for one round (run on the GPU box through gpurun; outputs land in gpurun_out/
# and are copied to profiles/ afterwards).  usage: tools/refresh_profiles.sh r01
set -o pipefail
TAG=${1:-r01}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "== bench lines"; date
python3 $R/bench.py | tail -1 > $O/${TAG}_bench_default_line.json || exit 1
python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 | tail -1 > $O/${TAG}_bench_cfg5_line.json || exit 1
echo "== kernel traces"; date
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_def -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/kt_def.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_cfg5 -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_cfg5.log 2>&1 || exit 1
python3 $R/tools/summarize_profile.py /tmp/kt_def > $O/${TAG}_bench_default_kernels.json
python3 $R/tools/summarize_profile.py /tmp/kt_cfg5 > $O/${TAG}_bench_cfg5_kernels.json
python3 $R/tools/small_system_gaps.py /tmp/kt_def 600 > $O/${TAG}_bench_default_gaps.txt
echo "== PMC passes (one counter per pass)"; date
for wl in cfg5 h2o_tz; do
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pmc_${wl}_$c -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/pmc_${wl}_$c.log 2>&1 || exit 1
    python3 $R/tools/summarize_profile.py /tmp/pmc_${wl}_$c > $O/${TAG}_pmc_${c}_${wl}.json
  done
  python3 $R/tools/pmc_bench_traffic.py $wl /tmp/pmc_${wl}_FETCH_SIZE /tmp/pmc_${wl}_WRITE_SIZE $O/${TAG}_traffic.json > /dev/null
done
date
echo done
