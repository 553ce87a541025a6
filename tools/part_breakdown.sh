# per-kernel GPU time of one interleaved row part of the bench frame (development aid)
# usage: tools/part_breakdown.sh <parts> <part> [reps]
parts=$1; part=$2; reps=${3:-6}
export TMPDIR=/tmp
out=gpurun_out/part_${parts}_${part}
rm -rf $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/part_trace.py $parts $part $reps > $out.log 2>&1 || exit 1
f=$(find $out -name "*kernel_stats.csv")
python3 - "$f" $reps <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); n=int(sys.argv[2])
for r in rows[:16]:
    print(r["Name"][:72].ljust(72), str(round(int(r["Calls"])/n,1)).rjust(6), "calls", str(round(float(r["TotalDurationNs"])/n/1e6,3)).rjust(9),"ms/frame")
print("sum ms/frame", round(sum(float(r["TotalDurationNs"]) for r in rows)/n/1e6,3))
PY
