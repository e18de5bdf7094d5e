#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv of bench.py and prints, for the last frames, the GPU idle time between consecutive
dispatches (frame boundary = k_resolve -> next k_trace_primary).  usage: frame_gaps.py <kernel_trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
for a, b in zip(idx[-6:-1], idx[-5:]):
    fr = rows[a + 1:b + 1]
    prev = int(rows[a]["End_Timestamp"])
    span = (int(fr[-1]["End_Timestamp"]) - prev) / 1000
    gaps, busy = [], 0.0
    for r in fr:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gaps.append(((s - prev) / 1000, r["Kernel_Name"][:28]))
        busy += (e - s) / 1000
        prev = e
    print("frame %.1f us busy %.1f us idle %.1f us :" % (span, busy, span - busy), " ".join("%s+%.1f" % (n, g) for g, n in gaps if g > 1.0))
