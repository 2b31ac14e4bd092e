#!/bin/bash
# prof_ao2mo.sh OUT -- rocprofv3 kernel trace of three AO->MO transforms at n = 220 (tools/first_call_ao2mo.py), condensed into OUT
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
D="$HERE/gpurun_out/prof_ao2mo"
rm -rf "$D"; mkdir -p "$D"
OUT="$(realpath -m "$1")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 "$HERE/tools/first_call_ao2mo.py" > "$D/run.log" 2>&1
python3 "$HERE/tools/summarize_profile.py" "$D" > "$OUT"
