#!/bin/bash
# Any-hit cell sweeps of the default bench frame under a few settings (pass trace on stderr).
for env in "$@"; do
  echo "== ${env:-default}"
  env $env RT_AMD_TRACE_PASSES=1 timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/probe.err >gpurun_out/probe.json
  grep "cell sweep:.*ms" gpurun_out/probe.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/probe.json').read().strip().splitlines()[-1])
print("frame ms", round(d['ms_per_step'],2), "Mrays/s", round(d['value'],1), "primary ms", round(d['roofline']['avg_launch_ms'],2), "shadow sweeps ms", round(d['roofline']['anyhit']['ms_per_frame'],2), "launches", d['roofline']['anyhit']['launches_per_frame'])
PY
done
