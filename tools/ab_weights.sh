#!/bin/bash
# A/B of builds and settings on one box: label, then env assignments for bench.py
cd "$(dirname "$0")/.."
one() { label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-28s %.3f ms/step kernel %.3f ms %.0f GB/s %s' % ('$label', d['ms_per_step'], r['kernel_ms'], r['achieved'], r.get('xcd_weights_32nds', '')))"
}
for rep in 1 2 3; do
  one "occ7 depth2" A=1
  one "occ8 depth2" GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_occ8.so
  one "occ8 depth1" GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_occ8.so GORT_EXPAND_DEPTH=1
  one "occ7 depth1" GORT_EXPAND_DEPTH=1
done
