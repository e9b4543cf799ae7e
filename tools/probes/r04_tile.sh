#!/bin/bash
# round 4: the consensus + residual tile route -- parity tests, the A/B over divergences, phase ablation, and a kernel trace.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_tile; mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_twist.py -x -q -m gpu -k "tile_kernel or few_assemblies or config3 or genomes" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
fi
timeout 900 python tools/probes/ab_tile_kernel.py > $O/ab.txt 2>&1
cat $O/ab.txt
[ -n "$AB_DBG_LIST" ] && AB_DBG=$AB_DBG_LIST timeout 600 python tools/probes/ab_tile_kernel.py 2>&1 | tee $O/ablation.txt
if [ -n "$TRACE" ]; then
(cd /tmp && AB_RATES=0.001,0.01,0.03 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_tile_prof -o tile -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_tile_kernel.py > $GRAFT_REPO_ROOT/$O/ab_prof.txt 2>&1)
find /tmp/r04_tile_prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find /tmp/r04_tile_prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
python3 tools/probes/r04_trace_calls.py $O/kernel_trace.csv | tee $O/trace_calls.txt
fi
