#!/bin/bash
# rocprofv3 passes over the closest-hit sweep (run on the GPU box from the repo root).
# Each pass is its own short run; counters are never combined with API tracing.
#   usage: tools/profile_sweep.sh <out-dir> [quick_sweep.py args...]
set -uo pipefail
out="$1"; shift
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@")
run() {  # name, rocprof args...
  local name="$1"; shift
  timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 tools/quick_sweep.py --reps 2 "${args[@]}" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run trace --kernel-trace --stats || exit 1
run pmc_issue --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run pmc_flops --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS || exit 1
run pmc_fetch --kernel-trace --pmc FETCH_SIZE || exit 1
run pmc_write --kernel-trace --pmc WRITE_SIZE || exit 1
find "$out" -name "*.csv" | head -40
