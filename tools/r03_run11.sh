#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rs -k "lut_alloc or rccl or bench" > $O/gpu_tests11.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests11.log | cut -c1-600
STEPS=40 timeout -k 10 900 python3 tools/scaling_estimate.py 2>&1 | grep -v amdgpu.ids | tee $O/scaling_estimate3.log
