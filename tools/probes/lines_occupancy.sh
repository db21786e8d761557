#!/bin/bash
# stream_lines_kernel at fewer resident waves per CU (a larger LDS allocation per wave, measuring build): how much of its time is
# occupancy?  160 KB per CU: 17 680 B (pitch 34) -> 9 waves, 20 480 -> 8, 23 400 -> 7, 27 300 -> 6, 32 768 -> 5, 40 960 -> 4
cd "$(dirname "$0")/../.."
export GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_ab.so
for nw in 100 190; do
  for b in 0 20480 23400 27300 32768 40960 54600; do
    echo "GORT_LINES_LDS_BYTES=$b"
    GORT_LINES_LDS_BYTES=$b python3 tools/bench_lines.py 1000000 $nw 15 2>/dev/null || exit 1
  done
done
