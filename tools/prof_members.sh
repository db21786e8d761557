#!/bin/bash
# The ensemble observation operator (gort_rsurf_members_stream_dev) under the profiler, per library: kernel trace, HBM traffic
# (WRITE_SIZE / FETCH_SIZE in separate passes), SQ counters.  gpurun: bash tools/prof_members.sh [lib ...] -> gpurun_out/pm/
set -u
R=$PWD; OUT=$R/gpurun_out/pm; rm -rf $OUT; mkdir -p $OUT
PY=$(python3 -c 'import sys;print(sys.executable)')
cd /tmp; export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES"
for lib in ${@:-gort_amd/libgort_amd.so}; do
  tag=$(basename $lib .so)
  export GORT_AMD_LIB=$R/$lib
  for shape in "1000 100" "1000 2101"; do
    set -- $shape; n=$1; nw=$2; t=${tag}_${n}x${nw}
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$t -- $PY $R/tools/probes/members_stream.py 1000 $n $nw > $OUT/kt_$t.log 2>&1
    f=$(find $OUT/kt_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kt_$t.csv
    for ctr in WRITE_SIZE FETCH_SIZE; do
      timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/${ctr}_$t -- $PY $R/tools/probes/members_stream.py 1000 $n $nw > $OUT/${ctr}_$t.log 2>&1
      python3 $R/tools/summarize_pmc.py $OUT/${ctr}_$t "" > $OUT/${ctr}_$t.json
    done
    timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1_$t -- $PY $R/tools/probes/members_stream.py 1000 $n $nw > $OUT/sq1_$t.log 2>&1
    python3 $R/tools/summarize_pmc.py $OUT/sq1_$t "" > $OUT/sq1_$t.json
    timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/sq2_$t -- $PY $R/tools/probes/members_stream.py 1000 $n $nw > $OUT/sq2_$t.log 2>&1
    python3 $R/tools/summarize_pmc.py $OUT/sq2_$t "" > $OUT/sq2_$t.json
  done
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python3 - $OUT <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "stream_lines" in k or "expand_flat_stream" in k or "geometry_stream" in k or "member_stream_bands" in k:
            print(os.path.basename(f), k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
