#!/bin/bash
# Kernel trace of the shared-sun-zenith stream (gort_stream_suns.hip): what the stages in front of the expansion cost, and the
# bytes the expansion writes.   tools/probes/suns_trace.sh [LINES]
cd "$(dirname "$0")/../.." || exit 1
R=$PWD; N=${1:-1000000}
PY=$(python3 -c 'import sys; print(sys.executable)')
export BENCH_STREAM_MODES=2
OUT=$R/gpurun_out/suns_trace; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $PY "$R/tools/bench_stream.py" $N 3 "91 sun" ) > "$OUT/stats.log" 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $PY "$R/tools/bench_stream.py" $N 1 "91 sun" ) > "$OUT/pmc_write.log" 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $PY "$R/tools/bench_stream.py" $N 1 "91 sun" ) > "$OUT/pmc_fetch.log" 2>&1
$PY - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/*/*_kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:16]:
        print("%-64s calls %5s avg %10.1f us" % (row["Name"][:64], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
for t in pmc_write pmc_fetch; do $PY tools/summarize_pmc.py "$OUT/$t" 2>/dev/null | head -40; done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
