#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for n in 65536 1048576; do
for w in 16808 33616; do for k in 16 32 64 128 100000; do
  echo -n "n=$n K=$k W=$w : "
  GORT_STREAM_GROUP=0 GORT_STREAM_STEPS=$k GORT_STREAM_WAVES=$w timeout -k 10 100 python3 tools/bench_stream.py $n 12 "all" 2>&1 | grep "grouping=1" | cut -c40-100
done; done; done; done
