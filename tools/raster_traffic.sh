#!/bin/bash
# HBM-side read and write counters of the primary sweep's launch in bench.py's default frame
# (GPU box, repo root; one counter per pass):  tools/raster_traffic.sh <out-dir>
set -uo pipefail
out="$1"; mkdir -p "$out"; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-general-mt > "$out/pmc_$c.log" 2>&1 || { echo "pass $c failed"; tail -3 "$out/pmc_$c.log"; exit 1; }
done
python3 tools/pmc_summary.py "$out" k_sweep_raster
