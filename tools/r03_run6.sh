#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pipe_and_cli.py tests/test_gpu_parity.py -x -q -m gpu -rs -k "pipe or scanf or numbers or mirror or c3 or fused or few_band" > $O/gpu_tests6.log 2>&1; echo "pytest rc=$?"; tail -6 $O/gpu_tests6.log | cut -c1-600
for r in 4 8; do echo "GORT_GEOM_ROWS=$r"; GORT_GEOM_ROWS=$r timeout -k 10 200 python3 tools/bench_configs.py 2>&1 | grep "^C3"; done | tee $O/c3_rows.log
GORT_COMMIT=$(cat gpurun_out/commit.txt 2>/dev/null) tools/profile_round3.sh 2>&1 | tail -40
