#!/bin/bash
# round 4: every profile the round's numbers are read from, in one call on the GPU box (repo root).
set -u
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r04_final; mkdir -p $O
export TMPDIR=/tmp
# 1. the tile route: A/B over divergences + phase clocks + per-call kernel times, then the MFMA / VALU pipe counters
SKIP_TESTS=1 TRACE=1 AB_DBG_LIST=16 bash tools/probes/r04_tile.sh > $O/tile_ab.log 2>&1
cp gpurun_out/r04_tile/ab.txt $O/tile_ab.txt; cp gpurun_out/r04_tile/ablation.txt $O/tile_phase_clocks.txt; cp gpurun_out/r04_tile/trace_calls.txt $O/tile_trace_calls.txt
cp gpurun_out/r04_tile/kernel_stats.csv $O/tile_kernel_stats.csv
bash tools/probes/r04_tile_pmc.sh gpurun_out/r04_tile_pmc 0.001 > $O/tile_pmc_0.1pct.txt 2>&1
bash tools/probes/r04_tile_pmc.sh gpurun_out/r04_tile_pmc1 0.01 > $O/tile_pmc_1pct.txt 2>&1
# 2. every CLI kernel: trace + FETCH_SIZE + WRITE_SIZE passes per section
bash tools/cli_kernels_profile.sh gpurun_out/r04_cli > $O/cli_profile.log 2>&1
# 3. SQ counters of the -L count kernel, the -s summary kernel, the partition histogram
bash tools/sq_counters.sh gpurun_out/r04_sq count_L summary_65 merged_hist > $O/sq.log 2>&1
# 3b. the classifier's summary kernel (four rows of a wavefront at a time) against round 3's, its phase ablation, kernel times, SQ counters;
#     the -L count kernel after the networks' exchanges moved to the vector pipe
for c in 65 64 10 100 130 200; do AB_CLASSES=$c AB_DBG=1 python tools/probes/ab_summary_wave.py; done > $O/summary_ab_new.txt 2>&1
KPOP_TUNE_DBG=$((1<<30)) AB_CLASSES=65,64,10,100,130,200 python tools/probes/ab_summary_wave.py > $O/summary_ab_round3_kernel.txt 2>&1
CLASSES="65 64 100 130" bash tools/probes/r04_summary.sh > $O/summary_kernels.txt 2>&1
bash tools/probes/r04_summary_pmc.sh > $O/summary_pmc.txt 2>&1
python tools/probes/probe_count.py > $O/count_wave_probe.txt 2>&1
# 4. the bench line, plain; then the whole line (config legs included) under the kernel trace
python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o bench -- python3 $R/bench.py --no-children --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/bench_trace
tail -c 300 $O/bench.err; ls $O
