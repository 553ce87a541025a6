#!/bin/bash
# Primary (pixel-grid) sweep time of one bench workload under forced launch shapes.
# usage: raster_probe.sh WORKLOAD "ENV=..." ...
wl="$1"; shift
for env in "$@"; do
  echo "== $wl ${env:-default}"
  env $env timeout -k 10 200 python -u bench.py --workload "$wl" --steps 3 --warmup 1 --no-cpu-baseline --no-general-mt 2>gpurun_out/probe.err >gpurun_out/probe.json || { tail -5 gpurun_out/probe.err; exit 1; }
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/probe.json').read().strip().splitlines()[-1])
r=d['roofline']
print("frame ms", round(d['ms_per_step'],2), "primary ms", round(r['avg_launch_ms'],3), r.get('kernel'), "shadow ms", round(r['anyhit']['ms_per_frame'],2))
PY
done
