#!/bin/bash
# HBM bytes of the few-band LUT forms (geometry_grid_kernel writing the samples itself) against the algorithmic 8 B per sample:
# separate rocprofv3 --pmc passes for WRITE_SIZE and FETCH_SIZE at 7, 16 and 100 bands, and the kernel's time from a trace.
cd "$(dirname "$0")/../.." || exit 1
R=$PWD; PY=$(python3 -c 'import sys; print(sys.executable)')
OUT=$R/gpurun_out/few_band_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
for nw in 7 16 100; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$nw" -- $PY "$R/tools/probes/mid_band_grid.py" $nw ) > "$OUT/stats_$nw.log" 2>&1
  for c in WRITE_SIZE FETCH_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/${c}_$nw" -- $PY "$R/tools/probes/mid_band_grid.py" $nw ) > "$OUT/${c}_$nw.log" 2>&1
  done
  $PY - <<PY
import csv, glob, json, subprocess, sys
nw = $nw
t = None
for f in glob.glob("$OUT/stats_%d/*/*_kernel_stats.csv" % nw):
    for row in csv.DictReader(open(f)):
        if "geometry_grid_kernel" in row["Name"]:
            t = float(row["AverageNs"]) / 1e3; calls = row["Calls"]
b = {}
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    r = subprocess.run([sys.executable, "tools/summarize_pmc.py", "$OUT/%s_%d" % (c, nw)], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout)
        for k, v in d.items():
            if "geometry_grid_kernel" in k:
                b[c] = v[c]
    except Exception as ex:
        b[c] = None
alg = 91 * 91 * 361 * nw * 8
print("hemisphere x %3d bands: geometry_grid_kernel avg %.1f us over %s launches; WRITE_SIZE %s KB, FETCH_SIZE %s KB per launch; algorithmic %.1f MB; (WRITE + 2 FETCH) x 1024 / algorithmic = %s"
      % (nw, t or -1, calls if t else "?", b.get("WRITE_SIZE"), b.get("FETCH_SIZE"), alg / 1e6,
         "%.4f" % (((b["WRITE_SIZE"] or 0) + 2 * (b["FETCH_SIZE"] or 0)) * 1024 / alg) if b.get("WRITE_SIZE") else "?"))
PY
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
