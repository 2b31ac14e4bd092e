#!/bin/bash
# iter_timeline.sh OUT [o v window_ms] -- kernel sequence of one CCSD iteration (default: config 5; rocprofv3 kernel trace of tools/prof_run.py) -> OUT
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
O="${2:-20}"; V="${3:-200}"; WIN="${4:-30.0}"
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt_iter
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_iter -- python3 "$HERE/tools/prof_run.py" --o "$O" --v "$V" --iters 3 --triples 0 > /tmp/kt_iter.log 2>&1 || { tail -5 /tmp/kt_iter.log; exit 1; }
python3 "$HERE/tools/iter_timeline.py" /tmp/kt_iter "$WIN" > "$OUT"
