#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/pmc_workload.py into profiles/<tag>_pmc.csv and
profiles/traffic.json (the per-launch HBM traffic bench.py reports as roofline.traffic).

Corrections applied exactly as MI355X_MICROARCH.md (HBM section) prescribes, after checking them on the
calibration kernels of the same run (8-byte-per-lane coalesced streams of a known byte count):
  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     [FETCH_SIZE reads 1/2 on gfx950; unit KiB]
"""
import collections
import csv
import glob
import json
import os
import sys


def main(pmc_dir, tag, n, L, k, D):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(pmc_dir, "*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    rows = []
    for (kn, c), v in sorted(agg.items()):
        if kn.startswith("kpop::"):
            rows.append((kn, c, len(v), max(v), sum(v) / len(v)))
    out = os.path.join(root, "profiles", tag + "_pmc.csv")
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,max_per_dispatch,mean_per_dispatch\n")
        for r in rows:
            f.write('"%s",%s,%d,%.6g,%.6g\n' % r)
    fused = [kn for (kn, c) in agg if "count_twist_wave_kernel" in kn]
    kn = max(fused, key=lambda x: max(agg[(x, "FETCH_SIZE")]))
    fetch = sorted(agg[(kn, "FETCH_SIZE")])[len(agg[(kn, "FETCH_SIZE")]) // 2]
    write = sorted(agg[(kn, "WRITE_SIZE")])[len(agg[(kn, "WRITE_SIZE")]) // 2]
    tpath = os.path.join(root, "profiles", "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj["count_twist_wave_kernel:n=%d,L=%d,k=%d,D=%d" % (n, L, k, D)] = {
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
        "source": "profiles/%s_pmc.csv" % tag}
    json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
    print(out, tj)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *[int(x) for x in sys.argv[3:7]])
