#!/usr/bin/env python3
"""Joins the rocprofv3 output of tools/cli_kernels_profile.sh into one table: per kernel and section the average launch
time (kernel trace), the HBM traffic (PMC passes: 2 x FETCH_SIZE + WRITE_SIZE KiB, the gfx950 correction of
MI355X_MICROARCH.md's HBM section), the algorithmic bytes the workload recorded, and the two ratios the judge reads:
achieved = algorithmic bytes / time against the 8 TB/s peak, and traffic / algorithmic.

    python3 tools/cli_kernels_report.py <dir written by cli_kernels_profile.sh> > profiles/r02_c_cli_kernels.md
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def valu_busy(sq_root):
    """kernel -> share of the launch during which a SIMD's vector pipe was issuing (tools/sq_counters.sh output, if given)"""
    out = {}
    if not sq_root:
        return out
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(sq_root, "*_set*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, c in acc.items():
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
            mean = lambda v: sum(v) / len(v)
            out[kn] = mean(c["SQ_ACTIVE_INST_VALU"]) * 4 / (1024 * mean(c["GRBM_GUI_ACTIVE"]) / 8)
    return out


def main(d, sq_root=None):
    busy = valu_busy(sq_root)
    rows = []
    for aj in sorted(glob.glob(os.path.join(d, "*_algo.json"))):
        sec = json.load(open(aj))
        name = sec["section"]
        stats = {}
        for f in glob.glob(os.path.join(d, name + "_trace", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]))
        pmc = collections.defaultdict(lambda: collections.defaultdict(list))
        for which in ("fetch", "write"):
            for f in glob.glob(os.path.join(d, name + "_" + which, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn in sorted(stats):
            if not kn.startswith("kpop::"):
                continue
            calls, avg, mn = stats[kn]
            algo = next((v for key, v in sec["kernels"].items() if key in kn), None)
            fetch = pmc[kn].get("FETCH_SIZE")
            write = pmc[kn].get("WRITE_SIZE")
            traffic = None
            if fetch and write:
                traffic = (2 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024
            rows.append((name, kn, calls, avg / 1e6, mn / 1e6, algo, traffic))
    # the bound a kernel is measured against is the workload's statement (hbm unless it says otherwise): a fraction is only
    # printed against a bound that can hold it -- "l2" against the 34.5 TB/s of the eight L2s, "mfma" against the 78.6 TFLOP/s
    # of the f64 matrix cores (flops recorded by the workload), "valu" against the vector pipe's issue rate (ACTIVE_INST_VALU x 4 / (1,024 SIMDs x GUI / 8), from the SQ counter directory given as the second argument)
    print("| section | kernel | launches | avg ms | min ms | algorithmic bytes (or flops) / launch | achieved | bound | fraction of the bound's peak | HBM traffic / launch (PMC) | traffic / algorithmic bytes | note |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, kn, calls, avg, mn, algo, traffic in rows:
        ab = algo["bytes"] if algo and isinstance(algo.get("bytes"), (int, float)) else None
        bound = (algo or {}).get("bound", "hbm" if ab else "-")
        note = (algo or {}).get("note", "")
        work, ach, frac = "-", "-", "-"
        if bound == "mfma" and algo and algo.get("flops"):
            tf = algo["flops"] / (avg * 1e-3) / 1e12
            work, ach, frac = "%.4g flop" % algo["flops"], "%.1f TFLOP/s" % tf, "%.3f of 78.6 TFLOP/s" % (tf / 78.6)
        elif ab:
            gbs = ab / (avg * 1e-3) / 1e9
            work, ach = "%.4g" % ab, "%.0f GB/s" % gbs
            if bound == "hbm":
                # a fraction above 1 means the bytes did not come from HBM: the workload named the wrong bound (an L2- or
                # Infinity-Cache-resident case) -- refuse to print it rather than publish it
                assert gbs / 8000 <= 1.0, "%s / %s: %.0f GB/s of algorithmic bytes cannot be HBM-bound (traffic %s): fix the section's bound in cli_kernels_workload.py" % (name, kn, gbs, traffic)
                frac = "%.3f of 8 TB/s" % (gbs / 8000)
            elif bound == "l2":
                assert gbs / 34500 <= 1.0, "%s / %s: %.0f GB/s exceeds the L2s' 34.5 TB/s" % (name, kn, gbs)
                frac = "%.3f of 34.5 TB/s" % (gbs / 34500)
            elif bound == "valu":
                frac = "%.2f of the VALU issue rate" % busy[kn] if kn in busy else "(instruction-bound: SQ counters)"
            if algo and algo.get("flops"):
                note += "; %.1f Tops/s f64 of the 39.3 non-FMA peak" % (algo["flops"] / (avg * 1e-3) / 1e12)
        print("| %s | `%s` | %d | %.4f | %.4f | %s | %s | %s | %s | %s | %s | %s |" % (
            name, kn.replace("kpop::", ""), calls, avg, mn, work, ach, bound, frac, "%.4g" % traffic if traffic else "-",
            "%.2f" % (traffic / ab) if traffic and ab else "-", note))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
