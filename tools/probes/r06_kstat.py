#!/usr/bin/env python3
"""prints the rows of a rocprofv3 kernel_stats csv whose kernel name holds argv[2]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("  %-60s calls %3s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
