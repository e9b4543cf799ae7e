#!/usr/bin/env bash
# Every section of tools/cli_kernels_workload.py under rocprofv3: a kernel trace with stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own (counters never share a run with a trace summary; MI355X_MICROARCH.md).
# usage (on the GPU box, from the repo root): tools/cli_kernels_profile.sh gpurun_out/r02_cli [section ...]
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SECTIONS=${*:-count_L twist_reads twist_genomes summary_65 summary_1M merged_hist merged_hist_genomes merged_hist_mutants merged_hist_k7 merged_sort genomes_L fused_genomes fused_genomes_1pct fused_genomes_stream fused_genomes_unrelated}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in $SECTIONS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${s}_trace" -o "$s" -- python3 "$ROOT/tools/cli_kernels_workload.py" --only "$s" --algo-json "$OUT/${s}_algo.json" > "$OUT/${s}.log" 2>&1 || echo "trace of $s failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${s}_fetch" -o "$s" -- python3 "$ROOT/tools/cli_kernels_workload.py" --only "$s" >> "$OUT/${s}.log" 2>&1 || echo "FETCH pass of $s failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${s}_write" -o "$s" -- python3 "$ROOT/tools/cli_kernels_workload.py" --only "$s" >> "$OUT/${s}.log" 2>&1 || echo "WRITE pass of $s failed"
  echo "$s profiled"
  # the traces are large; the report needs the stats and the counter files only
  find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
done
python3 "$ROOT/tools/cli_kernels_report.py" "$OUT" > "$OUT/report.md"
