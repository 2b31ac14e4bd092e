#!/bin/bash
# clock_probe.sh OUT -- shader clock, power and temperature of the GPU sampled every 0.5 s while bench.py runs its config-5 steps
# (rocm-smi read-only queries): is the fp64 MFMA peak of the data sheet (2.4 GHz) the clock the (T) launches actually run at?
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$(realpath -m "$1")"
python3 "$HERE/bench.py" --steps 12 --warmup 2 --no-cpu-baseline --no-extra --no-live-pmc > /tmp/clock_bench.json 2> /tmp/clock_bench.err &
BP=$!
: > "$OUT"
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  { date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction)" ; } | tr '\n' ' ' >> "$OUT"
  echo >> "$OUT"
  sleep 0.5
done
wait $BP
echo "bench line:" >> "$OUT"; tail -1 /tmp/clock_bench.json | cut -c1-400 >> "$OUT"
