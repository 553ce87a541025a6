#!/bin/bash
# issue, LDS and L2 counters of the sweeps with tile culling on, bench.py's default frame (GPU box, repo root)
#   usage: tools/profile_culled.sh <out-dir>
set -uo pipefail
out="$1"; mkdir -p "$out"; export TMPDIR=/tmp
export RT_AMD_CULL=1
run() { name="$1"; shift
  timeout -s KILL 170 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$out/$name.log" 2>&1 || { echo "pass $name failed"; exit 1; }; }
run pmc_issue --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
run pmc_mem --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum
run pmc_FETCH_SIZE --pmc FETCH_SIZE
python3 tools/pmc_summary.py "$out" "true>"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
f = glob.glob(root + "/pmc_issue/vm/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "true>" in row["Kernel_Name"]:
        d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
for k, v in d.items():
    print(f"{k}: {len(v)} launches, {sum(v):.2f} ms in all, longest {max(v):.2f} ms")
PY
