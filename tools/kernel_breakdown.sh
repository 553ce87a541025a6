# per-kernel GPU time per frame of a bench.py workload (development aid); usage: tools/kernel_breakdown.sh <workload> [steps]
w=$1; steps=${2:-20}
export TMPDIR=/tmp
out=gpurun_out/breakdown_$w
rm -rf $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --workload $w --no-cpu-baseline --steps $steps --warmup 2 > $out.log 2>&1 || exit 1
f=$(find $out -name "*kernel_stats.csv")
python3 - "$f" $((steps+2)) <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); n=int(sys.argv[2])
for r in rows[:14]:
    print(r["Name"][:72].ljust(72), str(round(int(r["Calls"])/n,1)).rjust(6), "calls", str(round(float(r["TotalDurationNs"])/n/1e6,3)).rjust(9),"ms/frame")
print("sum ms/frame", round(sum(float(r["TotalDurationNs"]) for r in rows)/n/1e6,3))
PY
grep -o "\"ms_per_step\": [0-9.]*" $out.log
