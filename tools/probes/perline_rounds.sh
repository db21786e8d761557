#!/bin/bash
# 65 536 lines are ONE panel of the per-line kernel: does it matter how its wave count relates to the 7168 wave slots of
# the device (7 per SIMD)?  W = m x 2101 waves, K = steps so that one panel covers the stream.
cd "$(dirname "$0")/../.."
for rep in 1 2; do for m in 3 4 5 6 7 8 10 12; do
  W=$((m * 2101)); K=$(( (1075713 + W - 1) / W ))
  echo -n "W=$W K=$K rounds=$(python3 -c "print('%.2f' % ($W / 7168))") : "
  GORT_STREAM_STEPS=$K GORT_STREAM_WAVES=$W timeout -k 10 100 python3 tools/bench_stream.py 65536 20 "all" 2>&1 | grep "grouping=0" | cut -c42-100 || exit 1
done; done
