#!/bin/bash
# fused_big_sweep.sh -- iteration time over AFESP_FUSED_BIG_FLOP (the size from which a product keeps its own tiled launch inside a levelled sequence)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for big in ${BIGS:-3e7 1e8 2e8 4e8}; do
  for s in "5 53" "7 21" "5 110" "7 80" "9 60" "10 100"; do
    echo -n "BIG=$big o,v=$s: "; AFESP_FUSED_BIG_FLOP=$big python tools/fused_probe.py $s | grep "fused:"
  done
  echo -n "BIG=$big spin-orbital: "; AFESP_FUSED_BIG_FLOP=$big python tools/so_time.py | grep "iteration 6"
done
