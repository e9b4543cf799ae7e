#!/usr/bin/env bash
# SQ counters of the kernels a section of tools/cli_kernels_workload.py launches: one rocprofv3 --pmc pass per counter
# set (counters never share a run with a trace summary; MI355X_MICROARCH.md), reduced to per-kernel means by
# tools/sq_counters_report.py.  What DESIGN.md quotes for count_wave_kernel (VALU-issue share) and the distance kernels
# (VALU / LDS busy, LDS bank conflicts).
# usage (GPU box, repo root): tools/sq_counters.sh gpurun_out/r03_sq count_L summary_65 ...
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM")
for s in "$@"; do
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${s}_set$i" -o "$s" -- python3 "$ROOT/tools/cli_kernels_workload.py" --only "$s" > "$OUT/${s}_set$i.log" 2>&1 || echo "set $i ($set) of $s failed"
    find "$OUT/${s}_set$i" -name "*kernel_trace.csv" -delete; find "$OUT/${s}_set$i" -name "*agent_info.csv" -delete
    i=$((i+1))
  done
  echo "$s: counters collected"
done
python3 "$ROOT/tools/sq_counters_report.py" "$OUT" > "$OUT/report.txt"
