# times bench.py's default frame with each library variant (development aid)
for n in default "$@"; do
  if [ $n = default ]; then lib=raytracer_amd/librt_amd.so; else lib=variants/librt_$n.so; fi
  echo "== $n"
  RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('frame ms', round(d['ms_per_step'],1), 'primary ms', round(d['roofline']['avg_launch_ms'],1), 'tests/s', '%.3e'%d['tests_per_s'], 'shadow ms', round(d['shadow_sweeps_rank0']['ms_per_frame'],1), 'shadow tests', '%.3e'%d['shadow_sweeps_rank0']['tests_per_frame'])
" || exit 1
done
