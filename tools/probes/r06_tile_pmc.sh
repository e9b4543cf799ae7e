#!/bin/bash
# round 6: HBM-side traffic, L2 hit rate and the MFMA-pipe counter of the pipelined tile kernel at several numbers of dimensions
# (tile_pipe.h WIDE): one rocprofv3 --pmc pass a counter group over tools/probes/r06_tile_dims.py (FETCH_SIZE and WRITE_SIZE in passes
# of their own, as MI355X_MICROARCH.md prescribes).  Kernel <false, true> is the product's, <true, true> the ablation build's
# (R06_DBG=5: neither MFMAs nor residual gather -- the members' row stream and the slots' traffic alone).
# usage (GPU box, repo root): [R06_DBG=5] bash tools/probes/r06_tile_pmc.sh gpurun_out/r06_pmc "12:256 10:1635" "FETCH_SIZE WRITE_SIZE TCC SQ"
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
CASES="${2:-12:64 12:256 10:1635}"
GROUPS_="${3:-FETCH_SIZE WRITE_SIZE TCC SQ}"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
export TMPDIR=/tmp
for C in $CASES; do
  T=$(echo $C | tr ':' '_')
  for G in $GROUPS_; do
    case $G in
      TCC) CTR="TCC_HIT_sum TCC_MISS_sum";;
      SQ) CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY";;
      *) CTR="$G";;
    esac
    (cd /tmp && R06_CASES=$C R06_NO_STREAM=1 R06_REPS=2 timeout 150 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT/${G}_$T" -o t -- python3 "$ROOT/tools/probes/r06_tile_dims.py" > "$OUT/${G}_$T.log" 2>&1)
  done
done
python3 - "$OUT" $CASES <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
lines = ["pipelined tile kernel; per launch (mean over the launches of a pass)"]
for case in sys.argv[2:]:
    t = case.replace(":", "_")
    rows = collections.defaultdict(dict)
    for f in glob.glob(out + "/*_%s/**/*counter_collection.csv" % t, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tile_pipe" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (kn, c), v in acc.items():
            rows[kn][c] = sum(v) / len(v)
    logs = glob.glob(out + "/*_%s.log" % t)
    log = [l for l in open(logs[0]).read().splitlines() if l.startswith("k=") or "dbg" in l] if logs else []
    lines.append("k:D = %s   %s" % (case, " | ".join(x.strip() for x in log)))
    for kn, row in sorted(rows.items()):
        lines.append("  %s" % kn)
        if "FETCH_SIZE" in row:  # KiB; x 2 on gfx950 (MI355X_MICROARCH.md: 128-B requests tallied at 64 B)
            lines.append("     HBM-side read %.2f GB (FETCH_SIZE x 2), written %.2f GB" % (row["FETCH_SIZE"] * 2 * 1024 / 1e9, row.get("WRITE_SIZE", 0) * 1024 / 1e9))
        if "TCC_HIT_sum" in row:
            lines.append("     L2 hits %.4g misses %.4g (hit rate %.3f)" % (row["TCC_HIT_sum"], row.get("TCC_MISS_sum", 0), row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0))))
        if "GRBM_GUI_ACTIVE" in row:
            cyc = row["GRBM_GUI_ACTIVE"] / 8.0
            lines.append("     cycles/launch %.4g  MFMA pipes busy %.3f  VALU pipes busy %.3f  waves waiting %.3f of wave-cycles"
                         % (cyc, row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc), row.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024.0 * cyc),
                            row.get("SQ_WAIT_ANY", 0) / max(1.0, row.get("SQ_WAVE_CYCLES", 0))))
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
