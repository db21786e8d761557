#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for n in 65536 1048576; do
for w in 2101 4202 33616; do for k in 32 64 128; do
  echo -n "n=$n K=$k W=$w : "
  GORT_STREAM_STEPS=$k GORT_STREAM_WAVES=$w timeout -k 10 100 python3 tools/bench_stream.py $n 15 "all" 2>&1 | grep "grouping=0" | cut -c42-100
done; done; done; done
