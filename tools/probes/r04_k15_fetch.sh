#!/bin/bash
# round 4: HBM bytes fetched (FETCH_SIZE x 2, gfx950) by the fused count->twist on small-dimension twisters: the rows and the name -> row index
cd /tmp && export TMPDIR=/tmp
AB_CASES=${CASES:-15:16} AB_UNROLLS=8 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/k15 -o k -- python3 $GRAFT_REPO_ROOT/tools/probes/ab_small_batch_unroll.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("/tmp/k15/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "count_twist_wave_kernel" in r["Kernel_Name"]:
        acc[(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))].append(float(r["Counter_Value"]))
for g,v in sorted(acc.items(), key=lambda x: float(x[0]) if x[0].replace('.','').isdigit() else 0):
    print("grid %s: launches %d, FETCH_SIZE x2 KiB->bytes: %.3f GB (median)" % (g, len(v), sorted(v)[len(v)//2]*2*1024/1e9))
PY
