#!/bin/bash
# rocprofv3 kernel trace of the stream bench (one case), summary under gpurun_out/prof_stream/.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
CASE=${1:-91}
OUT=$R/gpurun_out/prof_stream_$CASE
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/bench_stream.py" 65536 20 "$CASE" > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
cat "$OUT/stats.log" | tail -5
for f in "$OUT"/stats/*/*_kernel_stats.csv; do cp "$f" "$OUT/kernel_stats.csv"; head -12 "$f"; done
