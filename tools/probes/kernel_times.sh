#!/bin/bash
# per-kernel durations of a probe under rocprofv3 --kernel-trace --stats: tools/probes/kernel_times.sh TAG script.py [args ...] -> gpurun_out/kt/TAG.txt
R=$PWD; TAG=$1; shift
PY=$(python3 -c 'import sys;print(sys.executable)')
OUT=$R/gpurun_out/kt; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -- $PY "$R/$1" "${@:2}" ) > $OUT/$TAG.log 2>&1
python3 - $OUT/$TAG > $OUT/$TAG.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("gort::(anonymous namespace)::", "").replace("void ", "")
        print("%-70s calls %5s avg %10.1f us min %10.1f max %10.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $OUT/$TAG
cat $OUT/$TAG.txt
