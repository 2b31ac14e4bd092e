#!/bin/bash
# Regenerates everything under profiles/ for one round (run on the GPU box through gpurun; outputs land in gpurun_out/
# and are copied to profiles/ afterwards).  usage: tools/refresh_profiles.sh r02
# bench.py's default workload is config 5 (o=20, v=200); the H2O/cc-pVTZ shape is --workload h2o_tz.
set -o pipefail
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "== bench lines"; date
python3 $R/bench.py | tail -1 > $O/${TAG}_bench_cfg5_line.json || exit 1
python3 $R/bench.py --workload h2o_tz --steps 50 --warmup 45 --no-extra | tail -1 > $O/${TAG}_bench_h2o_tz_line.json || exit 1
echo "== kernel traces"; date
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_cfg5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-live-pmc > $O/kt_cfg5.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_h2o -- python3 $R/bench.py --workload h2o_tz --steps 50 --warmup 45 --no-cpu-baseline --no-extra --no-live-pmc --steps-only > $O/kt_h2o.log 2>&1 || exit 1
python3 $R/tools/summarize_profile.py /tmp/kt_cfg5 > $O/${TAG}_bench_cfg5_kernels.json
python3 $R/tools/summarize_profile.py /tmp/kt_h2o > $O/${TAG}_bench_h2o_tz_kernels.json
python3 $R/tools/small_system_gaps.py /tmp/kt_h2o 600 > $O/${TAG}_bench_h2o_tz_gaps.txt
echo "== timelines: small-system iteration (launch-fused / laned / graph-replayed), spin-orbital iteration, config-5 iteration"; date
bash $R/tools/small_timeline.sh $O/${TAG}_small_timeline_fused.txt > /dev/null
AFESP_FUSED=0 AFESP_NO_GRAPH=1 bash $R/tools/small_timeline.sh $O/${TAG}_small_timeline_laned.txt > /dev/null
AFESP_FUSED=0 bash $R/tools/small_timeline.sh $O/${TAG}_small_timeline_replayed.txt > /dev/null
bash $R/tools/so_timeline.sh $O/${TAG}_so_timeline.txt
bash $R/tools/iter_timeline.sh $O/${TAG}_iter_timeline.txt
echo "== PMC passes (one counter per pass)"; date
bash $R/tools/refresh_traffic.sh $TAG || exit 1
bash $R/tools/pmc_ao2mo_fock.sh $TAG || exit 1
date
echo done
