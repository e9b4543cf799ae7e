#!/bin/bash
# round 4: the merged histogram's partition path -- parity tests, then a kernel trace per kpop_tune("histlds") mode
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_hist; mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests/test_gpu_count.py -x -q -m gpu -k "merged" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
fi
for mode in 0 1 3; do
(cd /tmp && AB_MODE=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_hist_$mode -o h -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_hist_partition.py 2>/dev/null | grep histlds | tee -a $GRAFT_REPO_ROOT/$O/ab.txt)
find /tmp/r04_hist_$mode -name "*kernel_trace.csv" -exec cp {} $O/trace_$mode.csv \;
python3 - $O/trace_$mode.csv $mode <<'PY' | tee -a $O/kernels.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# three workloads x three calls each: group by order of appearance of the table-scan kernel that ends every call (scan_apply NonZero)
calls, cur = [], collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void kpop::", "").replace("kpop::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "synth" in n: continue
    cur[n] = cur.get(n, 0) + d
    if "scan_apply_kernel<kpop::NonZero" in n:
        calls.append(cur); cur = collections.OrderedDict()
for i in (2, 5, 8):
    if i < len(calls):
        c = calls[i]
        hist = {k: v for k, v in c.items() if "hist" in k or "__amd_rocclr_fillBuffer" in k}
        print("mode %s call %d: histogram kernels %.0f us: %s" % (sys.argv[2], i, sum(hist.values()), "  ".join("%s %.0f" % (k[:40], v) for k, v in hist.items())))
PY
done
