#!/bin/bash
# mid_sweep.sh -- CCSD iteration time of mid-size systems on the small-system paths (launch-fused / lanes, AFESP_SMALL_MAX raised) against
# the single-stream call-by-call path they take by default above o^2 v^2 = 2^20
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-10,100 12,120 14,140 16,160}; do
    set -- ${shape/,/ }
    for mode in "AFESP_SMALL_MAX=1e9 AFESP_FUSED=1" "AFESP_SMALL_MAX=1e9 AFESP_FUSED=0" "AFESP_SMALL_MAX=0"; do
        echo "== o=$1 v=$2 $mode"
        env $mode timeout -k 10 200 python tools/fused_probe.py child $1 $2 2>&1 | grep "^T" || echo "   failed"
    done
done
