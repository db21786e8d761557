#!/bin/bash
# The shared-sun-zenith stream (gort_stream_suns.hip) on the measuring build: resident waves per CU (LDS padding) - is the
# expansion bound by the latency of its per-step scalar record or by the store pattern?   tools/probes/suns_ab.sh [LINES]
cd "$(dirname "$0")/../.." || exit 1
N=${1:-1000000}
PY=$(python3 -c 'import sys; print(sys.executable)')
export GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_ab.so BENCH_STREAM_MODES=2
for lds in 0 20000 40000 80000; do
  echo "== GORT_SUNS_LDS=$lds (workgroups of 4 waves per CU: $([ $lds = 0 ] && echo 8 || echo $((160000 / lds))))"
  GORT_SUNS_LDS=$lds "$PY" tools/bench_stream.py $N 5 "91 sun" 2>/dev/null
done
