#!/usr/bin/env python3
"""Gaps on the tracker's pose chain from a rocprofv3 --kernel-trace CSV: for consecutive k_tp_hyp / k_tp_frame
dispatches, kernel durations and the idle time between the end of one and the start of the next.
usage: tail_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, statistics as st
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "k_tp_" in n or "k_ti_" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k_tp_hyp" if "k_tp_hyp" in n else "k_tp_frame" if "k_tp_frame" in n else "k_ti_lists" if "k_ti_lists" in n else "k_ti_resolve"))
rows.sort()
for chain in (("k_tp_hyp", "k_tp_frame"), ("k_ti_lists", "k_ti_resolve")):
    seq = [r for r in rows if r[2] in chain]
    dur = {k: [] for k in chain}; gap = {}
    for a, b in zip(seq, seq[1:]):
        dur[a[2]].append(a[1] - a[0])
        gap.setdefault(a[2] + "->" + b[2], []).append(b[0] - a[1])
    for k, v in dur.items():
        if v: print("%-14s n %5d  median %7.1f us  mean %7.1f us" % (k, len(v), st.median(v) / 1e3, st.mean(v) / 1e3))
    for k, v in gap.items():
        v = [x for x in v if x < 200000]
        if v: print("  gap %-26s n %5d  median %6.2f us  mean %6.2f us" % (k, len(v), st.median(v) / 1e3, st.mean(v) / 1e3))
    if seq: print("  chain span per frame: %.1f us" % ((seq[-1][1] - seq[0][0]) / 1e3 / (len(seq) / 2)))
# the long gaps of the pose chain: where (dispatch index) and how long, with what the other kernels did meanwhile
seq = [r for r in rows if r[2] in ("k_tp_hyp", "k_tp_frame")]
allk = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        allk.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
allk.sort()
t0 = seq[0][0]
for i, (a, b) in enumerate(zip(seq, seq[1:])):
    g = b[0] - a[1]
    if g > 50000:
        inside = {}
        for k in allk:
            if k[0] < b[0] and k[1] > a[1] and "k_tp_" not in k[2]:
                inside[k[2]] = inside.get(k[2], 0) + (min(k[1], b[0]) - max(k[0], a[1])) / 1e3
        top = sorted(inside.items(), key=lambda kv: -kv[1])[:5]
        print("gap %7.1f us after dispatch %5d (frame %d) at t=%.2f ms: %s" % (g / 1e3, i, i // 2, (a[1] - t0) / 1e6, ", ".join("%s %.0f" % kv for kv in top)))
# timeline of everything inside the first long gap
first = None
for a, b in zip(seq, seq[1:]):
    if b[0] - a[1] > 1000000: first = (a[1], b[0]); break
if first and len(sys.argv) > 2:
    print("kernels inside the first gap > 1 ms (start us after the gap opens, duration us):")
    for k in allk:
        if k[1] > first[0] and k[0] < first[1] and "k_tp_" not in k[2]:
            print("  %8.1f %7.1f  %s" % ((k[0] - first[0]) / 1e3, (k[1] - k[0]) / 1e3, k[2]))
