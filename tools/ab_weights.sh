#!/bin/bash
# A/B of XCD duty weights (32nds) on one box, against ab_old/ (a previous build) when present.
cd "$(dirname "$0")/.."
one() { # dir label env...
  d=$1; label=$2; shift 2
  (cd $d && env "$@" python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-26s %.3f ms/step kernel %.3f ms %.0f GB/s %s' % ('$label', d['ms_per_step'], r['kernel_ms'], r['achieved'], r.get('xcd_weights_32nds', '')))")
}
for rep in 1 2; do
  [ -d ab_old ] && one ab_old "OLD"
  one . "equal" GORT_XCD_CALIBRATE=0
  one . "calibrated"
  one . "even32 odd25" GORT_XCD_WEIGHTS=32,25,32,25,32,25,32,25
  one . "even32 odd27" GORT_XCD_WEIGHTS=32,27,32,27,32,27,32,27
  one . "even25 odd32" GORT_XCD_WEIGHTS=25,32,25,32,25,32,25,32
done
