#!/bin/bash
# pp_split_sweep.sh -- config-5 CCSD iteration time against the K slices of the two pair products of the pp-ladder (AFESP_PP_SPLIT)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for s in 0 2 3 4 5 8 13 16; do echo "AFESP_PP_SPLIT=$s"; AFESP_PP_SPLIT=$s python3 tools/prof_run.py --iters 5 --triples 0 --ladder 5 | tail -3; done
