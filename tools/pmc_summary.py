#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch) and
print/write a small summary; used on the GPU box so only the summary travels back."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(dirs, out):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if name.startswith("void "):
                        name = name[5:]
                    name = name.replace("(anonymous namespace)::", "").split("(")[0]
                    name = name.split("<")[0]          # template instances of one kernel under its name
                    if not name.startswith("k_"):
                        continue
                    c = agg[name][row["Counter_Name"]]
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
    summary = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} | {"_dispatches": max(v[1] for v in cs.values())}
               for k, cs in agg.items()}
    with open(out, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, cs in sorted(summary.items()):
        print(k, {c: round(v, 1) for c, v in cs.items()})


if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
