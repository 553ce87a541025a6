#!/bin/bash
# Frame time against the number of fine cells per ray of the cell sweep (GPU box, repo root):
#   tools/cells_fine_sweep.sh <out.txt>
set -uo pipefail
out="$1"; : > "$out"
for w in soup1m_4k mesh100k_4k bunny_class_4k soup100k_1080p; do
for f in 1 2 4 8 16; do
  RT_AMD_CELLS_FINE=$f timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-general-mt --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); a=d['roofline']['anyhit']
        print('$w fine/ray $f', 'frame', round(d['ms_per_step'],3), 'anyhit', round(a['ms_per_frame'],3), 'fine', a.get('fine_cells_per_cell'), 'lookups %.2e' % a.get('row_lookups_per_frame',0), 'rays1x1 %.2e' % a.get('ray_pairs_one_by_one_per_frame',0))
" >> "$out"
done; done
cat "$out"
