#!/bin/bash
# round 5: the merged (-l) histogram's kernels, call by call (rocprofv3 --kernel-trace), default mode and the old dense-table mode (histlds 4)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_hist; mkdir -p $O; rm -f $O/ab.txt $O/kernels.txt
export TMPDIR=/tmp
for mode in ${MODES:-1 4}; do
(cd /tmp && AB_MODE=$mode timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r05_hist_$mode -o h -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_hist_partition.py 2>/dev/null | grep histlds | tee -a $GRAFT_REPO_ROOT/$O/ab.txt)
find /tmp/r05_hist_$mode -name "*kernel_trace.csv" -exec cp {} $O/trace_$mode.csv \;
python3 - $O/trace_$mode.csv $mode <<'PY' | tee -a $O/kernels.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by gaps of more than 2 ms between kernels (host work between the calls of the probe)
calls, cur, last = [], [], None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void kpop::", "").replace("kpop::", "")
    if "synth" in n: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last is not None and s - last > 1_000_000 and cur:
        calls.append(cur); cur = []
    cur.append((n, (e - s) / 1e3, s, e)); last = e
if cur: calls.append(cur)
for ci, c in enumerate(calls):
    span = (c[-1][3] - c[0][2]) / 1e3
    print("mode %s call %d: %d kernels, busy %.0f us, first start to last end %.0f us: %s" % (sys.argv[2], ci, len(c), sum(x[1] for x in c), span,
          "  ".join("%s %.0f" % (x[0][:34], x[1]) for x in c)))
PY
done
