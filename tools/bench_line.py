"""Reads bench.py output on stdin and prints the main figures of its JSON line (development aid)."""
import json, sys
for l in sys.stdin:
    if not l.startswith('{"metric'):
        continue
    d = json.loads(l); r = d["roofline"]; a = r["anyhit"]
    print(d["config"]["workload"], "%.1f %s" % (d["value"], d["unit"]), "frame ms", round(d["ms_per_step"], 2),
          "primary ms", round(r["avg_launch_ms"], 2), "%.3e tests/s" % r["tests_per_s"],
          "ppl", r["sweep_kernel"]["pixels_per_lane"], "chunks", r["sweep_kernel"]["triangle_chunks"],
          "shadow ms", round(a["ms_per_frame"], 2), "launches", a["launches_per_frame"],
          "shadow tests %.3e" % a["tests_per_frame"])
