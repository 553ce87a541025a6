#!/bin/bash
# Counters of the cell-lattice any-hit sweeps (k_sweep_cells) inside bench.py's default frame
# (GPU box, repo root):  tools/profile_cells.sh <out-dir>
set -uo pipefail
out="$1"; mkdir -p "$out"; export TMPDIR=/tmp
run() { name="$1"; shift
  timeout -s KILL 170 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; }; }
run pmc_issue --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
run pmc_mem --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]; import os; PAT = os.environ.get("KPAT", "k_sweep_cells")
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if PAT in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", f.split("/")[-3])
    for d in sorted(per, key=int)[-2:]:         # the profiled frame's two launches
        print("  dispatch", d, " ".join(f"{k}={v:.4e}" for k, v in sorted(per[d].items())))
for f in sorted(glob.glob(root + "/pmc_issue/*/*_kernel_trace.csv")):
    for row in csv.DictReader(open(f)):
        if PAT in row["Kernel_Name"]:
            print("  trace", row["Dispatch_Id"], (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6, "ms")
PY
