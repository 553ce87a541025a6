#!/bin/bash
# instruction counters of the any-hit apex sweeps inside bench.py's default frame (GPU box, repo root)
#   usage: tools/profile_shadow.sh <out-dir>
set -uo pipefail
out="$1"; mkdir -p "$out"; export TMPDIR=/tmp
run() { name="$1"; shift
  timeout -s KILL 170 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$out/$name.log" 2>&1 || { echo "pass $name failed"; exit 1; }; }
run pmc_issue --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
run pmc_flops --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
root = sys.argv[1]
tot = collections.defaultdict(float); n = 0
for f in glob.glob(root + "/*/vm/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_sweep_apex" in row["Kernel_Name"] and "false>" in row["Kernel_Name"]:     # the all-pairs launches
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
dur = 0.0; launches = 0
f = glob.glob(root + "/pmc_issue/vm/*_kernel_trace.csv")[0]
for row in csv.DictReader(open(f)):
    if "k_sweep_apex" in row["Kernel_Name"] and "false>" in row["Kernel_Name"]:
        dur += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9; launches += 1
line = [l for l in open(root + "/pmc_issue.log") if l.startswith('{"metric')][0]
d = json.loads(line)
# bench.py renders the timed steps and one more frame with the library's profiling on: the
# counters cover all of them, the pair tests are per frame
per_frame = d["roofline"]["anyhit"]["tests_per_frame"]
frames = launches / d["roofline"]["anyhit"]["launches_per_frame"]
shadow_tests = per_frame * frames
print(f"k_sweep_apex: {launches} launches in {frames:g} frames, {dur*1e3:.1f} ms under the counters, {shadow_tests:.3e} pair tests")
for k in sorted(tot):
    print(f"   {k:26s} {tot[k]:.4e}   per pair (x64 lanes) {tot[k]*64/shadow_tests:.3f}")
PY
