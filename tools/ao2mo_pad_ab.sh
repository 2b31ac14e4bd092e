#!/bin/bash
# AO->MO + MP2 with the temporaries' columns padded to whole K steps (default) and not (AFESP_AO2MO_PAD=0), alternating, one process each:
# same-session A/B (boxes differ by several per cent).  usage: tools/ao2mo_pad_ab.sh [v ...]   (n = 20 + v)
HERE=$(cd "$(dirname "$0")/.." && pwd)
for v in ${@:-200}; do
  for rep in 1 2 3; do
    for pad in 1 0; do
      echo -n "n=$((v+20)) pad=$pad: "
      AFESP_AO2MO_PAD=$pad timeout -k 10 120 python3 "$HERE/tools/ao2mo_time.py" 20 $v 4 2>&1 | tail -1
    done
  done
done
