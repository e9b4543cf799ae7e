cd $GRAFT_REPO_ROOT
for chunk in 0 25600 51200 102400; do for depth in 4 8; do python tools/pipeline_probe.py --trace --outputs 3 --chunk $chunk --depth $depth --inflight 10 2>/dev/null | tail -1; done; done
for chunk in 0 25600 102400; do python tools/pipeline_probe.py --trace --outputs 2 --chunk $chunk --depth 4 --inflight 10 2>/dev/null | tail -1; done
