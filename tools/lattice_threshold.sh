#!/bin/bash
# Frame time of the small workloads with the cell-lattice any-hit sweep taking over at different
# phase sizes (RT_AMD_LATTICE_MIN_PAIRS = rays x triangles of a phase): GPU box, repo root.
#   usage: tools/lattice_threshold.sh <out.txt>
set -uo pipefail
out="$1"; : > "$out"
for w in teapot_seq_4k teapot_class_480p soup100k_1080p; do
  for mp in 2e10 2e9 2e8 1e7; do
    RT_AMD_LATTICE_MIN_PAIRS=$mp timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-general-mt --steps 10 --warmup 2 2> /dev/null \
      | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d = json.loads(l); r = d['roofline']['anyhit']
        print('$w', 'min_pairs', '$mp', 'frame ms', round(d['ms_per_step'], 3), 'shadow ms', round(r['ms_per_frame'], 3), 'launches', r['launches_per_frame'], 'lattice launches', r.get('lattice_launches'))
" >> "$out" || exit 1
  done
done
cat "$out"
