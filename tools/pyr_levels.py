#!/usr/bin/env python3
"""Per-level duration of k_pyr_level (and the other front-end kernels) from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys, collections, statistics as st
by = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (n, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0))
        by[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(by, key=lambda k: (k[0], -k[1] * max(k[2], 1))):
    if k[0].startswith("k_"):
        print("%-18s grid %8d x %5d x %4d  n %4d  median %8.1f us" % (k[0], k[1], k[2], k[3], len(by[k]), st.median(by[k]) / 1e3))
