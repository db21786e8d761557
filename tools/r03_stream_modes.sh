#!/bin/bash
# Round 3: where the LDS-resident stream kernel's time goes: the kernel, its arithmetic without stores, its stores with
# trivial arithmetic; task heights 8 and 16; against the flat-panel kernel on the same box.  Logs under gpurun_out/r03/.
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
L=$O/${1:-stream_modes}.log
run() { echo "== $*" | tee -a $L; timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $L; }
for n in 65536 1048576; do
  for mode in 0 1 2; do
    for k in 8 16; do
      echo "-- GORT_STREAM_LDS_MODE=$mode GORT_STREAM_LDS_STEPS=$k" | tee -a $L
      GORT_STREAM_LDS_MODE=$mode GORT_STREAM_LDS_STEPS=$k BENCH_STREAM_MODES=2 run python3 tools/bench_stream.py $n 20 "all distinct"
    done
  done
  BENCH_STREAM_MODES=1 run python3 tools/bench_stream.py $n 20 "all distinct"
done
