#!/bin/bash
# SQ counters of count_twist_tile_pipe_kernel on 5,000 wuhan mutants (one rocprofv3 --pmc pass per counter set; counters never
# share a run with a trace summary).  usage (GPU box, repo root): bash tools/probes/r05_pipe_pmc.sh gpurun_out/r05_pipe_pmc [rate]
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
RATE=${2:-0.001}
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC")
i=0
for set in "${SETS[@]}"; do
  PIPE_CASES=64:0.001 PIPE_TIME=1 PIPE_ONLY=pipe PIPE_RATES=$RATE rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/set$i" -o pipe -- python3 "$ROOT/tools/probes/pipe_check.py" > "$OUT/set$i.log" 2>&1 || echo "set $i ($set) failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "tile_pipe" in k or "tile_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, c in acc.items():
    lines.append(k)
    # keep the launches of the big batch only (the largest values of each counter: 5,000 sequences, not the 64 of the check)
    for name in sorted(c):
        v = sorted(c[name])[-6:]
        lines.append("    %-28s %14.6g   (mean of the %d largest launches)" % (name, sum(v) / len(v), len(v)))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
