#!/bin/bash
# round 4: the classifier's summary (65 classes x 100,000 rows x 64 dimensions) -- per-kernel durations of one call
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_summary; mkdir -p $O
export TMPDIR=/tmp
for c in ${CLASSES:-65 64}; do
(cd /tmp && AB_CLASSES=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_sum_$c -o s -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_summary_wave.py 2>/dev/null | grep classes)
find /tmp/r04_sum_$c -name "*kernel_stats.csv" -exec cp {} $O/stats_$c.csv \;
python3 - $O/stats_$c.csv <<'PY' | tee $O/kernels_$c.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void kpop::", "").replace("kpop::", "")
    if "at::" in n or "elementwise" in n: continue
    print("%-70s calls %4s  avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
