#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc output (one CSV row per dispatch and counter) per kernel.
   usage: summarize_pmc.py <dir with *_counter_collection.csv below it> [substring filter] -> JSON on stdout"""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gort"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (_, k), ctrs in per_dispatch.items():
        for c, v in ctrs.items():
            agg[k][c].append(v)
out = {}
for k, ctrs in agg.items():
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("gort::", "").split("(")[0]
    d = {"launches": max(len(v) for v in ctrs.values())}
    for c, v in ctrs.items():
        d[c] = sum(v) / len(v)
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
        wc = d["SQ_WAVE_CYCLES"]
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if c in d:
                d[c + "_share_of_wave_cycles"] = d[c] / wc
        if "SQ_WAVES" in d and d["SQ_WAVES"] > 0 and "SQ_INSTS_VALU" in d:
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
    if "GRBM_GUI_ACTIVE" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_ACTIVE_INST_VALU" in d:
        # gfx94x formula of VALUBusy (no gfx950 section in derived_counters.xml): quad-cycles x 4 over SIMDs x GPU-active cycles
        d["VALUBusy_pct"] = 100.0 * d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / d["GRBM_GUI_ACTIVE"]
    out[name] = d
json.dump(out, sys.stdout, indent=1, sort_keys=True)
