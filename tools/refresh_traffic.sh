#!/bin/bash
# The PMC part of refresh_profiles.sh alone (FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE / MFMA busy, one counter per pass).
set -o pipefail
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for wl in cfg5 h2o_tz; do
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pmc_${wl}_$c -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-live-pmc --steps-only > $O/pmc_${wl}_$c.log 2>&1 || exit 1
    python3 $R/tools/summarize_profile.py /tmp/pmc_${wl}_$c > $O/${TAG}_pmc_${c}_${wl}.json
  done
  python3 $R/tools/pmc_bench_traffic.py $wl /tmp/pmc_${wl}_FETCH_SIZE /tmp/pmc_${wl}_WRITE_SIZE $O/${TAG}_traffic.json > /dev/null
done
python3 $R/tools/pmc_mfma_busy.py $TAG $O
echo done
