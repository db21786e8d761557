#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rs -k "energy or albedo or c4 or mirror or c3 or fused" > $O/gpu_tests7.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests7.log | cut -c1-600
for r in 128 64 128 64; do echo -n "GORT_GEOM_THREADS=$r "; GORT_GEOM_THREADS=$r timeout -k 10 200 python3 tools/bench_configs.py 2>&1 | grep "^C3"; done | tee $O/c3_threads.log
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 2101 2>&1 | grep -v amdgpu.ids | tee $O/energy_stream3.log
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 7 2>&1 | grep -v amdgpu.ids | tee -a $O/energy_stream3.log
STEPS=40 timeout -k 10 900 python3 tools/scaling_estimate.py 2>&1 | grep -v amdgpu.ids | tee $O/scaling_estimate.log
