#!/bin/bash
# els_mgpu.sh N [AFESP_COMM] -- run els_amd as N ranks (one process per GPU) in the current directory (els.in, *.dat).
# Rank 0 prints the reference's output; the (T) triples are split over the ranks.
# AFESP_COMM: rccl (default, N <= number of GPUs) or host (ranks may share a GPU: rehearsal on a one-GPU box).
# A rank that fails ends the job: the others are terminated (a rank left alone in ncclCommInitRank or ncclAllReduce would wait
# for ever), the exit status is the failing rank's.  AFESP_JOB_TIMEOUT (seconds, default 86400) bounds the whole job.
set -u
N=${1:?usage: els_mgpu.sh N [rccl|host]}
COMM=${2:-rccl}
LIMIT=${AFESP_JOB_TIMEOUT:-86400}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
JOB="$(mktemp -d "${TMPDIR:-/tmp}/afesp_job.XXXXXX")"
export AFESP_WORLD=$N AFESP_COMM=$COMM AFESP_COMM_FILE="$JOB/bootstrap"
pids=()
AFESP_RANK=0 "$HERE/els_amd" &          # rank 0 owns this script's stdout / stderr
pids+=($!)
for ((r = 1; r < N; r++)); do
    AFESP_RANK=$r "$HERE/els_amd" > "$JOB/rank$r.out" 2> "$JOB/rank$r.err" &
    pids+=($!)
done
sleep "$LIMIT" < /dev/null > /dev/null 2>&1 &     # the job's time limit: a child like the ranks, waited for with them
watchdog=$!
stop_all() {
    kill -TERM "${pids[@]}" 2>/dev/null
    sleep 2
    kill -KILL "${pids[@]}" 2>/dev/null
}
trap 'stop_all; kill "$watchdog" 2>/dev/null; exit 130' INT TERM
rc=0
left=$N
while [ "$left" -gt 0 ]; do
    done_pid=0
    wait -n -p done_pid "${pids[@]}" "$watchdog"
    code=$?
    if [ "$done_pid" = "$watchdog" ]; then
        rc=124
        echo "els_mgpu.sh: time limit of $LIMIT s reached: ending the ranks; rank outputs are under $JOB" >&2
        stop_all
        break
    fi
    left=$((left - 1))
    if [ "$code" -ne 0 ]; then
        rc=$code
        echo "els_mgpu.sh: a rank failed (exit status $code): ending the other ranks; rank outputs are under $JOB" >&2
        stop_all
        break
    fi
done
kill "$watchdog" 2>/dev/null
wait 2>/dev/null
[ "$rc" -eq 0 ] && rm -rf "$JOB"
exit "$rc"
