#!/bin/bash
# Everything the committed profiles of this round are made from (GPU box, repo root), each step
# under its own time limit; stops at the first failure.
#   usage: tools/round3_profiles.sh <out-dir>        (a directory under gpurun_out/)
set -uo pipefail
out="$1"; mkdir -p "$out"; export TMPDIR=/tmp
step() { echo "== $*"; }

step "issue rates of the sweeps' instruction mixes"
timeout -k 10 200 tools/microbench/valu_rate > "$out/valu_rate.txt" 2>&1 || exit 1
grep "raster.*_box" "$out/valu_rate.txt"

step "kernel trace + HBM counters of the default bench run"
bash tools/profile_bench.sh "$out/bench" > "$out/bench_run.log" 2>&1 || { tail -5 "$out/bench_run.log"; exit 1; }
tail -12 "$out/bench_run.log"

step "issue counters of the primary sweep"
KPAT=k_sweep_raster_wave bash tools/profile_cells.sh "$out/raster" > "$out/raster_counters.txt" 2>&1 || exit 1
tail -8 "$out/raster_counters.txt"

step "issue counters of the cell sweeps"
bash tools/profile_cells.sh "$out/cells" > "$out/cells_counters.txt" 2>&1 || exit 1
tail -8 "$out/cells_counters.txt"

step "the general Moller-Trumbore sweep: executed fp32 instructions by kind"
for w in mesh100k_4k soup1m_4k; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d "$out/general_$w" -- python3 tools/general_mt.py $w > "$out/general_$w.json" 2> "$out/general_$w.err" || { tail -3 "$out/general_$w.err"; exit 1; }
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/general_trace_$w" -- python3 tools/general_mt.py $w > "$out/general_trace_$w.json" 2> "$out/general_trace_$w.err" || exit 1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
root = sys.argv[1]
for w in ("mesh100k_4k", "soup1m_4k"):
    line = json.loads([l for l in open(f"{root}/general_{w}.json") if l.startswith("{")][-1])
    f = glob.glob(f"{root}/general_{w}/*/*_counter_collection.csv")[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if "k_sweep<" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    d = per[sorted(per, key=int)[-1]]           # the last launch of the program: a timed one
    flop = 64 * (d["SQ_INSTS_VALU_ADD_F32"] + d["SQ_INSTS_VALU_MUL_F32"] + 2 * d["SQ_INSTS_VALU_FMA_F32"])
    pairs = line["tests"]
    print(f"{w}: k_sweep pairs {pairs:.4e}  ADD {d['SQ_INSTS_VALU_ADD_F32']:.4e} MUL {d['SQ_INSTS_VALU_MUL_F32']:.4e} "
          f"FMA {d['SQ_INSTS_VALU_FMA_F32']:.4e} VALU {d['SQ_INSTS_VALU']:.4e} wave-instr -> {flop:.4e} flop = {flop / pairs:.2f} per pair "
          f"(bench: {line['flop_per_test']} per pair, {line['ms']:.1f} ms, frac {line['frac']:.3f})")
PY

step "strong-scaling prediction and one eighth's timeline"
bash tools/band_round3.sh "$out/band" final || exit 1

step "the other workloads"
bash tools/bench_others.sh "$out/other_workloads.jsonl" || exit 1
echo "all steps done"
