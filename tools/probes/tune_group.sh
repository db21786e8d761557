#!/bin/bash
# Distance between the eight XCD write windows (mode 3): G consecutive panels per XCD and group.
# usage: tools/tune_group.sh "<group list>" steps waves [nsza] [nslab]
export PROBE_FILL=0 PROBE_ROUNDS=1 PROBE_NSZA=${4:-91} PROBE_N=${5:-4}
for g in $1; do
  echo "== group=$g steps=$2 waves=$3"
  GORT_EXPAND_GROUP=$g GORT_EXPAND_STEPS=$2 GORT_EXPAND_WAVES=$3 GORT_EXPAND_XCD=${XCDMODE:-3} timeout -k 10 120 python3 tools/placement_probe2.py 2>&1 | grep -E 'round' | sed 's/torch.*//'
done
