#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -rs > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -12 $O/gpu_tests.log | cut -c1-400
timeout -k 10 200 python3 tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.log
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 2101 2>&1 | grep -v amdgpu.ids | tee $O/energy_stream2.log
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 7 2>&1 | grep -v amdgpu.ids | tee -a $O/energy_stream2.log
