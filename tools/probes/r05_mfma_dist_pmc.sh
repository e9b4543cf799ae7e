# MFMA-pipe and VALU counters of the large-reference summary's kernels (one --pmc pass: never with a trace summary).
# usage (GPU box, repo root): bash tools/probes/r05_mfma_dist_pmc.sh gpurun_out/r05_mfma_dist
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc" -o d -- python3 "$ROOT/tools/probes/r05_mfma_dist.py" > "$OUT/pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:60]
    if "kpop" not in r["Kernel_Name"]: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0) or 1
    # SQ_* are summed over the XCDs' SEs; MFMA busy is per-SIMD cycles x 4 SIMDs ... normalise as tools/probes/mfma_counters.sh does
    print("%-60s n=%3d GUI/launch %10.0f  MFMA pipes busy %.3f  VALU active %.3f" % (
        k, n[k], gui / n[k] / 8, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024), c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (gui / 8 * 1024)))
PY
