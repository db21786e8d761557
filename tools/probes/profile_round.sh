#!/bin/bash
# Profiles of the headline benchmark for profiles/rNN/ (run on the GPU box through gpurun):
#   1. bench.py unprofiled (the number of record)
#   2. rocprofv3 --kernel-trace --stats of the same command   -> per-kernel average durations
#   3. rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE, one pass each (TCC slots: MI355X_MICROARCH.md)
# Outputs under gpurun_out/prof/.  The python program itself follows `--` (no env/bash hop under rocprofv3).
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R" && timeout -k 10 300 python3 bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1; echo "pmc write rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
cd "$R" && python3 - <<'PY'
import csv, glob, collections, json, os
out = "gpurun_out/prof"
summary = {}
for tag, ctr in (("pmc_write", "WRITE_SIZE"), ("pmc_fetch", "FETCH_SIZE")):
    files = glob.glob("%s/%s/*/*_counter_collection.csv" % (out, tag))
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "gort" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    summary[ctr] = {k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in agg.items()}
json.dump(summary, open(out + "/pmc_hbm_traffic.json", "w"), indent=1)
for f in glob.glob(out + "/stats/*/*_kernel_stats.csv"):
    print(open(f).read()[:1500])
for ctr, d in summary.items():
    for k, v in d.items():
        if "expand_flat" in k:
            print(ctr, "expand_flat_kernel: %.3f GB per launch over %d launches" % (v["mean_KB"] * 1024 / 1e9, v["launches"]))
for name in ("bench_unprofiled.json", "stats.log"):
    for l in open(os.path.join(out, name)):
        if l.startswith("{"):
            d = json.loads(l)
            print(name, "ms_per_step %.3f kernel_ms %.3f achieved %.1f GB/s value %.4e" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["value"]))
PY
