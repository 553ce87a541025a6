# average shader clock of every any-hit launch of one row part (development aid):
#   tools/part_clock.sh <parts> <part>
parts=$1; part=$2
export TMPDIR=/tmp
out=gpurun_out/clock_${parts}_${part}
rm -rf $out
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/part_trace.py $parts $part 3 > $out.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv, glob, sys
root = sys.argv[1]
dur = {}
for r in csv.DictReader(open(glob.glob(root + "/*/*_kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for r in csv.DictReader(open(glob.glob(root + "/*/*_counter_collection.csv")[0])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        name, ns = dur[r["Dispatch_Id"]]
        if "k_sweep" in name:
            rows.append((int(r["Dispatch_Id"]), name.split("::")[-1][:22], ns, float(r["Counter_Value"])))
rows.sort()
for d, name, ns, cyc in rows[-14:]:
    print(f"{name:24s} {ns/1e6:8.3f} ms  {cyc/8/ns:5.2f} GHz")
PY
