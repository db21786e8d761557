#!/usr/bin/env python3
"""Gaps between consecutive launches of one kernel in a rocprofv3 --kernel-trace CSV, and what ran in between.
   usage: trace_gaps.py <kernel_trace.csv> <kernel name substring>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
needle = sys.argv[2]
prev_end, between = None, []
gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gort::(anonymous namespace)::", "")
    if needle in r["Kernel_Name"]:
        if prev_end is not None:
            gaps.append(((s - prev_end) / 1e3, (e - s) / 1e3, list(between)))
        prev_end, between = e, []
    elif prev_end is not None:
        between.append("%s %.1f..%.1f" % (name, (s - prev_end) / 1e3, (e - prev_end) / 1e3))
for g in gaps[3:12]:
    print("gap %.1f us before a launch of %.1f us; in between (us after the previous end): %s" % g)
import statistics
print("median gap %.1f us over %d launches" % (statistics.median(x[0] for x in gaps), len(gaps)))
