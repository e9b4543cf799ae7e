#!/bin/bash
# round 6: per-kernel times of the large-reference summary (rocprofv3 --kernel-trace --stats), one run a (rows, mode).
# usage (GPU box, repo root): bash tools/probes/r06_summary_trace.sh gpurun_out/sumtrace "1024 256" "1 2"
set -u
OUT="$(cd "$(dirname "$1")" 2>/dev/null && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
export TMPDIR=/tmp
for rows in ${2:-1024}; do for mode in ${3:-1}; do
  (cd /tmp && R06_D_CASES= R06_S_CASES=${R06_R1:-1000000}:$rows:${R06_DIMS:-64} R06_S_MODES=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r${rows}_m$mode" -o t -- python3 "$ROOT/tools/probes/r06_dist_dims.py" > "$OUT/r${rows}_m$mode.log" 2>&1)
  echo "== rows $rows mode $mode"; grep -h "summary_mfma\|^-s" "$OUT/r${rows}_m$mode.log"
  f=$(find "$OUT/r${rows}_m$mode" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-70s calls %4s  avg %9.1f us  total %9.1f us  %5s %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
done; done
