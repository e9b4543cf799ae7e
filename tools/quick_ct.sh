#!/bin/bash
# development shortcut: rebuild ONE object of kpop_amd/csrc (default count_twist) and relink, leaving the others as they are
# (every object depends on every header in the Makefile: a touched header rebuilds all thirteen, three minutes)
cd "$(dirname "$0")/../kpop_amd/csrc" || exit 1
obj=${1:-count_twist}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter -c $obj.hip -o $obj.o 2>&1 | grep -E "error|Error" -A6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkpop_hip.so runtime.o pipeline.o multi.o packed.o twister.o count_twist.o twist_dense.o sort_count.o distance.o distance_mfma.o summary_large.o splits.o ca.o counter.o && touch *.o ../libkpop_hip.so
