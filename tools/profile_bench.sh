#!/bin/bash
# rocprofv3 passes over bench.py's default run (GPU box, repo root): a kernel trace with
# statistics, then one counter pass each for the HBM-side read and write byte counters.
# Counter passes never carry API tracing.
#   usage: tools/profile_bench.sh <out-dir> [bench.py args...]
set -uo pipefail
out="$1"; shift
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1 || { echo "trace pass failed"; exit 1; }
grep '^{"metric' "$out/bench.log" > "$out/bench_line.json"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/pmc_$c.log" 2>&1 || { echo "pass $c failed"; exit 1; }
done
python3 tools/pmc_summary.py "$out" k_sweep > "$out/pmc_summary.txt"
cat "$out/pmc_summary.txt"
