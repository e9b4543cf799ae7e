#!/bin/bash
# build everything here (hipcc cross-compiles), then run a command on the GPU box: tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.." || exit 1
make -C kpop_amd/csrc -j4 2>&1 | grep -E "error|warning: unused|Error" ; make -C kpop_amd/host -j4 2>&1 | grep -E "error|Error"; make -C oracle 2>&1 | grep -E "error|Error"
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
