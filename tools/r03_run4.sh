#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rs -k "energy or albedo or c4 or bench or rccl" > $O/test_energy_bench.log 2>&1; tail -15 $O/test_energy_bench.log
for mode in 0 3; do for n in 65536 1048576; do
  echo "-- GORT_STREAM_LDS_MODE=$mode" | tee -a $O/stream_modes3.log
  GORT_STREAM_LDS_MODE=$mode BENCH_STREAM_MODES=2 timeout -k 10 200 python3 tools/bench_stream.py $n 20 "all distinct" 2>&1 | grep -v amdgpu.ids | tee -a $O/stream_modes3.log
done; done
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 2101 2>&1 | grep -v amdgpu.ids | tee $O/energy_stream.log
timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 7 2>&1 | grep -v amdgpu.ids | tee -a $O/energy_stream.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1.json; tail -5 $O/bench_n1.err
