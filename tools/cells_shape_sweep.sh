#!/bin/bash
# Frame and shadow-sweep time of a workload with the cell sweep's shape and grid forced (GPU box, repo root):
#   tools/cells_shape_sweep.sh <workload> <out.txt>
set -uo pipefail
w="$1"; out="$2"; : > "$out"
run() {
  env "$@" timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-general-mt --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); a=d['roofline']['anyhit']
        print('$*', 'frame', round(d['ms_per_step'],3), 'anyhit', round(a['ms_per_frame'],3), 'cells/lane', a.get('cells_per_lane'), 'fine', a.get('fine_cells_per_cell'), 'entries %.2e' % a.get('triangles_entering_rare_path_per_frame',0), 'rays1x1 %.2e' % a.get('ray_pairs_one_by_one_per_frame',0))
" >> "$out"
}
run X=0
for r in 512 128 32 8; do run RT_AMD_CELLS_FORCE=$r; done
for c in 0.03 0.1 0.5; do run RT_AMD_CELLS_PER_RAY=$c; done
for c in 0.03 0.1; do run RT_AMD_CELLS_PER_RAY=$c RT_AMD_CELLS_FORCE=128; done
for f in 0.25 4; do run RT_AMD_CELLS_FINE=$f; done
cat "$out"
