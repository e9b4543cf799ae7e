# MFMA-pipe counters of the kernels that run on the matrix cores: one rocprofv3 --pmc pass (counters never share a run with a
# trace summary), reduced per kernel.  usage (GPU box, repo root): bash tools/probes/mfma_counters.sh gpurun_out/r03_mfma
set -u
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
AB_ONLY_GENOMES=1 AB_KS=7,8 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/dense" -o dense -- python3 "$ROOT/tools/ab_dense_twist.py" > "$OUT/dense.log" 2>&1
AB_GENOMES=5000 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/tile" -o tile -- python3 "$ROOT/tools/probes/ab_tile_kernel.py" > "$OUT/tile.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sec in ("dense", "tile"):
    f = glob.glob(out + "/" + sec + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(sec, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "mfma" in k or "dense" in k or "tile_kernel" in k or "counts_kernel" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    for k, c in acc.items():
        d = max(n[k], 1)
        mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d, c.get("GRBM_GUI_ACTIVE", 0) / d
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs: active cycles of the launch = gui / 8; 1,024 SIMDs each with one MFMA pipe
        print("%-44s launches %3d  SQ_VALU_MFMA_BUSY_CYCLES %.4g  GRBM_GUI_ACTIVE %.4g (sum of 8 XCDs)  -> MFMA pipes busy %.3f of the launch"
              % (k[:44], d, mf, gui, mf / (1024.0 * gui / 8.0) if gui else 0))
PY
