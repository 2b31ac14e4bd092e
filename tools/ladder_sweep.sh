#!/bin/bash
# ladder_sweep.sh -- the two pair products of the config-5 pp-ladder with tile codes / K slices of their own (AFESP_PP_TILES)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for t in "0,0,0,0,0,0" "0,0,0,4,2,4" "0,0,0,4,2,3" "0,0,0,4,2,5" "0,0,0,4,2,6" "4,2,4,4,2,4" "4,2,4,0,0,0" "4,2,5,4,2,4" "0,0,0,4,2,8" "0,0,0,2,4,4"; do
  echo -n "AFESP_PP_TILES=$t: "; AFESP_PP_TILES=$t python3 tools/prof_run.py --iters 2 --triples 0 --ladder 8 | tail -1
done
