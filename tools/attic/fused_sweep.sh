#!/bin/bash
# fused_sweep.sh -- iteration time of the launch-fused path (o=5, v=53) over the compile-time knobs of csrc/fused.hip
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for mm in 32 64 128 256 512; do for nb in 4 2; do for items in 2048 4096; do
  echo -n "MAX_MFMA=$mm NB=$nb ITEMS=$items: "
  AFESP_FUSED_MAX_MFMA=$mm AFESP_FUSED_NB=$nb AFESP_FUSED_ITEMS=$items python tools/fused_probe.py 5 53 | grep "fused:" 
done; done; done
