#!/bin/bash
# Strong-scaling prediction on one GPU and the kernel timeline of one eighth (GPU box, repo root):
#   tools/band_round3.sh <out-dir> <tag>
set -uo pipefail
out="$1"; tag="$2"; mkdir -p "$out"; export TMPDIR=/tmp
timeout -k 10 300 python -u tools/band_scaling.py --detail > "$out/$tag.txt" 2>&1 || { tail -5 "$out/$tag.txt"; exit 1; }
grep -v "^   part" "$out/$tag.txt"
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$tag" -- python3 tools/part_trace.py 8 3 4 16 > "$out/trace_$tag.log" 2>&1 || { tail -5 "$out/trace_$tag.log"; exit 1; }
f=$(find "$out/trace_$tag" -name "*kernel_trace.csv" | head -1)
python3 tools/frame_timeline.py "$f" > "$out/timeline_part3of8_$tag.txt"
tail -1 "$out/timeline_part3of8_$tag.txt"
