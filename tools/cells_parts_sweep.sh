for w in bunny_class_4k mesh100k_4k soup100k_1080p soup1m_4k; do
for p in 1 2 4 ""; do
  RT_AMD_CELLS_PARTS=$p timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-general-mt --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); a=d['roofline']['anyhit']
        print('$w parts=[$p]', 'frame', round(d['ms_per_step'],3), 'anyhit', round(a['ms_per_frame'],3))
"
done; done
