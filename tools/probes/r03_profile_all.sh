set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_q; mkdir -p $O
cd $R
python tools/pipeline_probe.py --timeline > $O/pipeline_timeline.txt 2>&1
bash tools/cli_kernels_profile.sh gpurun_out/r03_cli > $O/cli_profile.log 2>&1
python tools/ab_dense_twist.py > $O/dense_twist_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
AB_ONLY_GENOMES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_trace -o dense -- python3 $R/tools/ab_dense_twist.py > $O/dense_trace.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err; ls $O
