#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mirror or c3 or fused or few_band or lut_kernel or grid or c5 or members or ensemble or bench_json" > $O/gpu_tests18.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests18.log | cut -c1-500
timeout -k 10 200 python3 tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs_final.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rows6.json 2> $O/bench_rows6.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_rows6.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['first_draw']['value'], d['config5']['ms'])"
