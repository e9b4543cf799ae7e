#!/bin/bash
# round 6: everything profiles/r06_* is made of, on the round's last build.  usage (GPU box, repo root): bash tools/probes/r06_profiles.sh gpurun_out/r06_prof
set -u
OUT="$(cd "$(dirname "$1")" 2>/dev/null && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
# 1. the headline alone under rocprofv3 --kernel-trace --stats
bash tools/headline_kernel_stats.sh "$OUT/headline" > "$OUT/headline.txt" 2>&1
# 2. the tile route by dimensions: times, phase clocks, ablations (5,000 sequences), full size (50,000)
R06_CLOCKS=1 R06_DBG=1,4,5 timeout 600 python3 tools/probes/r06_tile_dims.py > "$OUT/tile_dims_5000.txt" 2>&1
R06_N=50000 R06_REPS=3 timeout 900 python3 tools/probes/r06_tile_dims.py > "$OUT/tile_dims_50000.txt" 2>&1
# 3. ... its counters (MFMA pipes busy; HBM-side bytes), one rocprofv3 --pmc pass a group
bash tools/probes/r06_tile_pmc.sh "$OUT/tile_pmc" "12:64 12:256 10:1635" "SQ FETCH_SIZE WRITE_SIZE" > "$OUT/tile_pmc.txt" 2>&1
# 4. distances on the matrix cores: times, and the MFMA-pipe counter of the kernels
timeout 900 python3 tools/probes/r06_dist_dims.py > "$OUT/dist_dims.txt" 2>&1
(cd /tmp && R06_D_CASES=1636:100000:1635 R06_S_CASES=650000:256:1635 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/dist_sq" -o t -- python3 "$ROOT/tools/probes/r06_dist_dims.py" > "$OUT/dist_sq.log" 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/dist_sq/**/*counter_collection.csv", recursive=True)
lines = []
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "mfma" in kn:
            acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, c in sorted(acc.items()):
        g = lambda nm: sum(c.get(nm, [0])) / max(1, len(c.get(nm, [0])))
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        if cyc:
            lines.append("%-60s launches %3d  cycles/launch %.4g  MFMA pipes busy %.3f  VALU pipes busy %.3f" % (kn[:60], len(c["GRBM_GUI_ACTIVE"]), cyc, g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), g("SQ_ACTIVE_INST_VALU") * 4 / (1024.0 * cyc)))
open(out + "/dist_sq.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
# 5. the driver-form line
timeout 900 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 300 "$OUT/bench.json" | head -c 300; echo
