#!/bin/bash
# els_mgpu.sh N [AFESP_COMM] -- run els_amd as N ranks (one process per GPU) in the current directory (els.in, *.dat).
# Rank 0 prints the reference's output; the (T) triples are split over the ranks.
# AFESP_COMM: rccl (default, N <= number of GPUs) or host (ranks may share a GPU: rehearsal on a one-GPU box).
set -u
N=${1:?usage: els_mgpu.sh N [rccl|host]}
COMM=${2:-rccl}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
JOB="$(mktemp -d "${TMPDIR:-/tmp}/afesp_job.XXXXXX")"
export AFESP_WORLD=$N AFESP_COMM=$COMM AFESP_COMM_FILE="$JOB/bootstrap"
pids=()
for ((r = 1; r < N; r++)); do
    AFESP_RANK=$r "$HERE/els_amd" > "$JOB/rank$r.out" 2> "$JOB/rank$r.err" &
    pids+=($!)
done
AFESP_RANK=0 "$HERE/els_amd"
rc=$?
for p in "${pids[@]}"; do
    wait "$p" || { rc=$?; echo "els_mgpu.sh: a rank failed; its output is under $JOB" >&2; }
done
[ $rc -eq 0 ] && rm -rf "$JOB"
exit $rc
