#!/bin/bash
# pp_sym_sweep.sh -- CCSD iteration time with the pair form of the pp-ladder / of t2 <ef|ia> forced on and off (AFESP_PP_SYM) at mid sizes
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-8,80 10,100 12,120 14,140 16,160 18,180}; do
    set -- ${shape/,/ }
    for mode in "AFESP_PP_SYM=0" "AFESP_PP_SYM=1"; do
        echo "== o=$1 v=$2 $mode"
        env $mode timeout -k 10 200 python tools/fused_probe.py child $1 $2 2>&1 | grep "^T unrep" || echo "   failed"
    done
done
