#!/bin/bash
# SQ counters of the stream kernels per band count (gpurun: bash tools/prof_lines.sh [bands ...]); summaries -> gpurun_out/pl/
set -u
R=$PWD; OUT=$R/gpurun_out/pl; rm -rf $OUT; mkdir -p $OUT
PY=$(python3 -c 'import sys;print(sys.executable)')
cd /tmp; export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for nw in ${@:-7 17 100 255}; do
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1_$nw -- $PY $R/tools/bench_lines.py 1000000 $nw 3 > $OUT/sq1_$nw.log 2>&1
  python3 $R/tools/summarize_pmc.py $OUT/sq1_$nw "" > $OUT/sq1_$nw.json
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python3 - $OUT <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/sq1_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "stream" in k or "geometry" in k:
            print(os.path.basename(f), k, "VALU/wave %.0f" % v.get("valu_insts_per_wave", 0), "waves %d" % v["SQ_WAVES"],
                  "valu share %.3f wait_any %.3f wait_inst %.3f" % (v.get("SQ_ACTIVE_INST_VALU_share_of_wave_cycles", 0), v.get("SQ_WAIT_ANY_share_of_wave_cycles", 0), v.get("SQ_WAIT_INST_ANY_share_of_wave_cycles", 0)),
                  "avg waves/SIMD %.2f" % (v["SQ_WAVE_CYCLES"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)), "clock-cycles %.0f" % (v["GRBM_GUI_ACTIVE"] / 8))
PY
