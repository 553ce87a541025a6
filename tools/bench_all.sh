# one summary line per bench.py workload (development aid); usage: tools/bench_all.sh [workload...]
ws="$@"; [ -z "$ws" ] && ws="soup1m_4k mesh100k_4k bunny_class_4k soup100k_1080p teapot_seq_4k teapot_class_480p"
for w in $ws; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline 2> gpurun_out/bench_$w.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['config']['workload'], 'frame ms', round(d['ms_per_step'],2), 'fps', round(d['frames_per_s'],2), 'Mrays/s', round(d['value'],1), 'primary tests/s %.3e'%r['tests_per_s'], 'frac', round(r['frac'],3), 'ppl', r['sweep_kernel']['pixels_per_lane'])
" || exit 1
done
