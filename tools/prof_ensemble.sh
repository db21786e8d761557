#!/bin/bash
# rocprofv3 kernel trace of the C5 ensemble setup + LUT chunks (tools/bench_ensemble.py): where do the setup ms go?
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_ensemble
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R" && timeout -k 10 300 python3 tools/bench_ensemble.py 1000 100 > "$OUT/unprofiled.log" 2>&1; cat "$OUT/unprofiled.log" | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/bench_ensemble.py" 1000 100 > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
for f in "$OUT"/stats/*/*_kernel_stats.csv; do cp "$f" "$OUT/ensemble_setup_stats.csv"; done
python3 - "$OUT/ensemble_setup_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-45s calls %4s total %10.1f us avg %10.1f us' % (r['Name'].split('(')[0].split('::')[-1][:45], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
