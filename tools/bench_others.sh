# bench.py's JSON line for every workload but the default one, one per line (development aid)
#   usage: tools/bench_others.sh <out.jsonl>
out=$1; : > "$out"
for w in mesh100k_4k bunny_class_4k soup100k_1080p teapot_seq_4k teapot_class_480p; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/bench_$w.err | grep '^{"metric' >> "$out" || exit 1
done
python3 - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"], "frame ms", round(d["ms_per_step"], 2), "primary", round(r["avg_launch_ms"], 2),
          "%.3e tests/s" % r["tests_per_s"], "issue", round(r["valu_issue"]["frac"], 3), "ppl", r["sweep_kernel"]["pixels_per_lane"],
          "chunks", r["sweep_kernel"]["triangle_chunks"], "shadow", round(r["anyhit"]["ms_per_frame"], 2), "launches", r["anyhit"]["launches_per_frame"])
PY
