#!/bin/bash
# Per-range counters of the any-hit sweeps (k_sweep_apex) of one interleaved row part of the bench
# frame (development aid; GPU box, repo root):  tools/apex_breakdown.sh <parts> <part> <out-dir>
# One rocprofv3 counter pass over tools/part_ranges.py, whose stderr carries the pair tests of every
# range; tools/apex_breakdown.py joins the two.
set -uo pipefail
parts=$1; part=$2; out=$3
export TMPDIR=/tmp
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_${parts}_${part}" -- python3 tools/part_ranges.py $parts $part > "$out/ranges_${parts}_${part}.log" 2>&1 || { echo "counter pass failed"; tail -5 "$out/ranges_${parts}_${part}.log"; exit 1; }
python3 tools/apex_breakdown.py "$out/pmc_${parts}_${part}" "$out/ranges_${parts}_${part}.log" > "$out/breakdown_${parts}_${part}.txt"
cat "$out/breakdown_${parts}_${part}.txt"
