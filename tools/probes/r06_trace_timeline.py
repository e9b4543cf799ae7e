#!/usr/bin/env python3
"""prints the last N kernel launches of a rocprofv3 --kernel-trace csv as a timeline (start, end in us from the first of them, stream/queue)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-60:]))
