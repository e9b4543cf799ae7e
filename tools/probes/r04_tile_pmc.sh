#!/bin/bash
# MFMA-pipe and VALU counters of the tile route's kernels (one rocprofv3 --pmc pass; counters never share a run with a trace summary)
# usage (GPU box, repo root): bash tools/probes/r04_tile_pmc.sh gpurun_out/r04_tile_pmc [rates]
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
RATES=${2:-0.001,0.01}
cd /tmp && export TMPDIR=/tmp
AB_RATES=$RATES AB_NO_UNRELATED=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/tile" -o tile -- python3 "$ROOT/tools/probes/ab_tile_kernel.py" > "$OUT/tile.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/tile/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if "tile_kernel" in k or "residual" in k or "stream_kernel" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
lines = []
for k, c in acc.items():
    d = max(n[k], 1)
    g = lambda nm: c.get(nm, 0) / d
    gui = g("GRBM_GUI_ACTIVE")
    # GRBM_GUI_ACTIVE comes summed over the 8 XCDs: active cycles of the launch = gui / 8; 1,024 SIMDs, one MFMA pipe and one VALU each;
    # SQ_ACTIVE_INST_VALU counts quad-cycles (x 4 = cycles)
    cyc = gui / 8.0
    lines.append("%-44s launches %3d  cycles/launch %.4g  MFMA pipes busy %.3f  VALU pipes busy %.3f  waves waiting %.3f of wave-cycles"
                 % (k[:44], d, cyc, g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0, g("SQ_ACTIVE_INST_VALU") * 4 / (1024.0 * cyc) if cyc else 0,
                    g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
