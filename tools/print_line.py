import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(round(d["ms_per_step"],3), "ms; primary", round(r["avg_launch_ms"],3), r["sweep_kernel"], "%.3g" % r["tests_per_s"], "shadow", round(r["anyhit"]["ms_per_frame"],3))
