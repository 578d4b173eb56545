#!/usr/bin/env python3
"""Print the k_* rows of a rocprofv3 kernel_stats.csv (found under argv[1]) compactly."""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if n.startswith("void "):
            n = n[5:]
        n = n.replace("(anonymous namespace)::", "").split("(")[0]
        n = n.split("<")[0]          # template instances of one kernel are reported under its name
        if n.startswith("k_"):
            print("%-20s calls %6s avg_us %9.1f min %8.1f max %8.1f total_ms %8.2f" % (
                n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
    if len(sys.argv) > 2:
        import shutil
        shutil.copy(path, sys.argv[2])
