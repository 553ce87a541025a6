# times bench.py's default frame with an environment variable set to each given value (development aid)
# usage: tools/env_ab.sh VAR value [value ...]
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('frame ms', round(d['ms_per_step'],1), 'primary ms', round(d['roofline']['avg_launch_ms'],1), 'shadow ms', round(d['shadow_sweeps_rank0']['ms_per_frame'],1), 'shadow tests', '%.3e'%d['shadow_sweeps_rank0']['tests_per_frame'], 'launches', d['shadow_sweeps_rank0']['launches_per_frame'])
" || exit 1
done
