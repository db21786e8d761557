#!/bin/bash
# Panel-shape sweep of expand_flat_kernel over several output allocations held at once.
# usage: tools/tune_panels.sh "<steps list>" "<waves list>" "<xcd list>" [nsza] [nslab]
export PROBE_FILL=0 PROBE_ROUNDS=1 PROBE_NSZA=${4:-91} PROBE_N=${5:-4}
for x in $3; do for w in $2; do for k in $1; do
  echo "== steps=$k waves=$w xcd=$x"
  GORT_EXPAND_STEPS=$k GORT_EXPAND_WAVES=$w GORT_EXPAND_XCD=$x timeout -k 10 120 python3 tools/placement_probe2.py 2>&1 | grep -E 'round|mapping' | sed 's/torch.*//'
done; done; done
