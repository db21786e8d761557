#!/bin/bash
# Round 3: the LDS-resident stream kernel against the flat-panel kernel on ONE box (A/B in one process per size),
# with the task height K and the XCD split of the LDS form swept.  Logs under gpurun_out/r03/.
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
run() { echo "== $*" | tee -a $O/stream_ab.log; timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/stream_ab.log; }
for n in 65536 1048576; do
  run python3 tools/bench_stream.py $n 20 "91 sun,all distinct"
done
for k in 4 8 16 32; do
  for n in 65536 1048576; do
    echo "-- GORT_STREAM_LDS_STEPS=$k" | tee -a $O/stream_ab.log
    GORT_STREAM_LDS_STEPS=$k BENCH_STREAM_MODES=2 run python3 tools/bench_stream.py $n 20 "all distinct"
  done
done
echo "-- GORT_STREAM_LDS_SPLIT=0" | tee -a $O/stream_ab.log
for n in 65536 1048576; do
  GORT_STREAM_LDS_SPLIT=0 BENCH_STREAM_MODES=2 run python3 tools/bench_stream.py $n 20 "all distinct"
done
