#!/usr/bin/env python3
"""per-call kernel durations (us) of the count->twist route from a rocprofv3 kernel trace CSV: one line per kpop_dev_count_twist call"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = ("count_twist_wave", "tile_kernel", "residual_kernel", "todo_kernel", "stream_kernel", "combine")
cur, calls = {}, []
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    n = r["Kernel_Name"]
    for key in keys:
        if key in n:
            cur[key] = cur.get(key, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "combine" in n:
        calls.append(cur)
        cur = {}
for j, c in enumerate(calls):
    if j % 8 == 7:
        print("call %3d  " % j + "  ".join("%s %7.0f" % (k, c[k]) for k in keys if k in c))
