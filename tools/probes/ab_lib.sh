#!/bin/bash
# interleaved A/B of two library builds on the per-line stream kernel: tools/probes/ab_lib.sh <suffix of variant>
cd "$(dirname "$0")/../.."
V=${1:-simple}
for rep in 1 2 3; do for lib in gort_amd/libgort_amd.so gort_amd/libgort_amd_$V.so; do
  for n in 65536 1048576; do
  echo -n "$lib n=$n: "; GORT_AMD_LIB=$PWD/$lib GORT_STREAM_GROUP=0 timeout -k 10 100 python3 tools/bench_stream.py $n 15 "all" 2>&1 | grep "grouping=1" | cut -c40-100
done; done; done
