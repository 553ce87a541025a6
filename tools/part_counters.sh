# issue counters of the sweep launches of one interleaved row part of the bench frame (development aid)
#   tools/part_counters.sh <parts> <part>
parts=$1; part=$2
export TMPDIR=/tmp
out=gpurun_out/pc_${parts}_${part}
rm -rf $out
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_issue -- python3 tools/part_trace.py $parts $part 2 > $out.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out k_sweep
python3 - $out <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*/*_kernel_trace.csv")[0])):
    if "k_sweep" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("::")[-1][:20]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(k, len(v), "launches, mean", sum(v) / len(v) / 1e6, "ms")
PY
