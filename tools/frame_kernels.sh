#!/bin/bash
# Kernel timeline of one timed frame of bench.py's default run (GPU box, repo root): tools/frame_kernels.sh <out-dir> [workload]
set -uo pipefail
out="$1"; wl="${2:-soup1m_4k}"; mkdir -p "$out"; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 bench.py --workload "$wl" --steps 3 --warmup 1 --no-cpu-baseline --no-general-mt > "$out/bench.log" 2>&1 || { tail -3 "$out/bench.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
tr = [i for i, r in enumerate(rows) if "k_transform" in r["Kernel_Name"]]
i0, i1 = tr[2], tr[3]          # the second timed step
t0 = int(rows[i0]["Start_Timestamp"]); end = 0
acc = {}
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    if "rocprim" in n: n = "rocprim"
    acc[n] = acc.get(n, 0) + (e - s)
    end = max(end, e)
    if e - s > 30000: print(f"{(s-t0)/1e6:8.3f} {(e-s)/1e3:8.1f}us q{r['Queue_Id']} {n}")
print("frame", (end - t0) / 1e6, "ms;", i1 - i0, "launches")
for k, v in sorted(acc.items(), key=lambda x: -x[1])[:16]: print(f"  {v/1e3:8.1f} us  {k}")
PY
