#!/bin/bash
# interleaved bench runs on one box (A/B aid): prints step time, LUT kernel time and the rest
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f ms/step  kernel %.3f ms  other %.3f ms  %.0f GB/s  value %.4e' % (d['ms_per_step'], r['kernel_ms'], d['ms_per_step']-r['kernel_ms'], r['achieved'], d['value']))"
done
