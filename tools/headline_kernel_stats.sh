#!/bin/bash
# The headline line alone under rocprofv3 --kernel-trace --stats: the dominant kernel's 23 launches (3 warm-up + 20 timed) and the
# distance kernels beside them, nothing else -- so that roofline.achieved can be recomputed from the CSV's average duration.
# usage (GPU box, repo root): bash tools/headline_kernel_stats.sh gpurun_out/r05_headline   ->  <out>/kernel_stats.csv, <out>/line.json
set -u
OUT="$(cd "$(dirname "$1")" 2>/dev/null && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o h -- python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline --no-children > "$OUT/line.json" 2> "$OUT/stderr.txt"
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
line = json.loads(open(out + "/line.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
k = [r for r in rows if "count_twist_wave_kernel" in r["Name"]][0]
avg_ms = float(k["AverageNs"]) / 1e6
alg = line["roofline"]["algorithmic_bytes_per_launch"]
print("count_twist_wave_kernel: %s launches, average %.4f ms in the CSV; the line's avg_launch_ms %.4f (HIP events); "
      "algorithmic %.0f bytes -> %.0f GB/s = %.3f of 8 TB/s from the CSV, %.3f in the line"
      % (k["Calls"], avg_ms, line["roofline"]["avg_launch_ms"], alg, alg / (avg_ms * 1e-3) / 1e9, alg / (avg_ms * 1e-3) / 1e9 / 8000.0, line["roofline"]["frac"]))
PY
