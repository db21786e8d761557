#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "pytest rc=$?"; tail -6 $O/gpu_tests_final.log | cut -c1-500
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -c 600 $O/bench_final.json
timeout -k 10 400 tools/cli_energy_throughput.sh 1000000 2>&1 | grep -v amdgpu.ids | tee $O/cli_energy_throughput.log
