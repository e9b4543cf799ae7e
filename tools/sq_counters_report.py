#!/usr/bin/env python3
"""Per-kernel means of the SQ counters collected by tools/sq_counters.sh, and the ratios DESIGN.md quotes:
  valu_issue = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-per-SIMD ...  (the counters are summed over the chip's SQs; ratios of
  two counters from the same kernel are what is comparable)."""
import collections
import csv
import glob
import os
import sys


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "*_set*", "**", "*_counter_collection.csv"), recursive=True)):
        sec = os.path.basename(f).split("_counter_collection")[0]
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "kpop::" not in kn:
                continue
            agg[(sec, kn)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (sec, kn), c in sorted(agg.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        print("%s  %s  (%d dispatches)" % (sec, kn, max(len(v) for v in c.values())))
        for k in sorted(m):
            print("    %-24s %16.6g" % (k, m[k]))

        def ratio(a, b):
            return m[a] / m[b] if a in m and b in m and m[b] else None
        out = []
        for label, a, b in (("VALU busy share of wave cycles", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
                            ("LDS busy share of wave cycles", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"),
                            ("VMEM busy share of wave cycles", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES"),
                            ("waiting share of wave cycles", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                            ("VALU instructions per wave", "SQ_INSTS_VALU", "SQ_WAVES"),
                            ("LDS instructions per wave", "SQ_INSTS_LDS", "SQ_WAVES"),
                            ("LDS bank-conflict cycles / LDS active cycles", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                            ("SQ busy cycles / GRBM active cycles", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")):
            r = ratio(a, b)
            if r is not None:
                out.append("    -> %-48s %.4g" % (label, r))
        print("\n".join(out))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
