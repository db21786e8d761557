#!/usr/bin/env python3
"""The launches of ONE kernel in a rocprofv3 --kernel-trace CSV, the warm-up launches dropped: bench.py runs blocks of
`warmup` untimed + `steps` timed launches of the dominant kernel (once on the plain first allocation, once on the headline's
buffer), and rocprofv3's own --stats summary averages over all of them, first touches included.  This prints the same
statistics over the timed launches only, block by block - directly comparable with the bench line's `kernel_ms`.
   usage: timed_launches.py <kernel_trace.csv> <kernel name substring> <warmup> <steps>   -> JSON on stdout"""
import csv, json, sys

path, needle, warmup, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]      # ms
blocks, i = [], 0
while i + warmup + steps <= len(dur):
    timed = dur[i + warmup: i + warmup + steps]
    blocks.append({"launches": len(timed), "avg_ms": sum(timed) / len(timed), "min_ms": min(timed), "max_ms": max(timed),
                   "warmup_launches_dropped_ms": dur[i: i + warmup]})
    i += warmup + steps
json.dump({"kernel": needle, "all_launches": len(dur), "all_avg_ms": sum(dur) / max(len(dur), 1), "warmup": warmup, "steps": steps,
           "blocks": blocks, "what": "block 0 = the plain first allocation (first_draw), last block = the headline's buffer"},
          sys.stdout, indent=1)
