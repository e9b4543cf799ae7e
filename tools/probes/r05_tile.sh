#!/bin/bash
# round 5: the pipelined tile kernel (tile_pipe.h) -- parity on small batches, ms per call against round 4's kernel and the
# streaming kernel at four divergences, the phase clocks of a producer and a consumer wavefront, and (PMC=1) the MFMA / VALU pipe
# counters of one rocprofv3 --pmc pass.  usage (GPU box, repo root): [PMC=1] bash tools/probes/r05_tile.sh gpurun_out/r05_tile
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
PIPE_NS=1 PIPE_TIME=1 PIPE_RATES=0.001,0.003,0.01,0.03 PIPE_COUNTS=1 timeout 600 python3 tools/probes/pipe_check.py > "$OUT/ab.txt" 2>&1
PIPE_CASES=64:0.001 PIPE_TIME=1 PIPE_ONLY=pipe PIPE_STAMPS=1 PIPE_RATES=0.001 timeout 300 python3 tools/probes/pipe_check.py 2>&1 | grep "wavefront" > "$OUT/stamps_0.1pct.txt"
PIPE_CASES=64:0.001 PIPE_TIME=1 PIPE_ONLY=pipe PIPE_STAMPS=1 PIPE_RATES=0.01 timeout 300 python3 tools/probes/pipe_check.py 2>&1 | grep "wavefront" > "$OUT/stamps_1pct.txt"
cat "$OUT/ab.txt" "$OUT/stamps_0.1pct.txt" "$OUT/stamps_1pct.txt"
if [ -n "${PMC:-}" ]; then
  for RATE in 0.001 0.01; do
    (cd /tmp && PIPE_CASES=64:0.001 PIPE_TIME=1 PIPE_RATES=$RATE rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/pmc_$RATE" -o tile -- python3 "$ROOT/tools/probes/pipe_check.py" > "$OUT/pmc_$RATE.log" 2>&1)
  done
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
lines = []
for rate in ("0.001", "0.01"):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % rate, recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "tile_pipe" in k or "tile_kernel" in k or "stream_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append("5,000 wuhan mutants at %.1f %%:" % (100 * float(rate)))
    for k, c in acc.items():
        # the launches of the 5,000-sequence batch only: the largest GRBM_GUI_ACTIVE values (the 64-sequence check runs first)
        n = max(1, len(c["GRBM_GUI_ACTIVE"]) // 2)
        g = lambda nm: sum(sorted(c.get(nm, [0]))[-n:]) / n
        cyc = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
        lines.append("  %-44s launches %3d  cycles/launch %.4g  MFMA pipes busy %.3f  VALU pipes busy %.3f  waves waiting %.3f of wave-cycles"
                     % (k[:44], n, cyc, g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0, g("SQ_ACTIVE_INST_VALU") * 4 / (1024.0 * cyc) if cyc else 0,
                        g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0))
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
fi
