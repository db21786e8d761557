#!/bin/bash
# per-line stream kernel on small streams: which wave count?  (stage time from HIP events)
cd "$(dirname "$0")/../.."
for n in 3000 8192 16384 32768; do
for w in 2101 4202 8404 16808 33616; do
  echo -n "n=$n W=$w : "
  GORT_STREAM_WAVES=$w timeout -k 10 100 python3 tools/bench_stream.py $n 30 "all" 2>&1 | grep "grouping=0" | cut -c42-130
done; done
