set -u
R=$PWD; OUT=$R/gpurun_out/pl; rm -rf $OUT; mkdir -p $OUT
PY=$(python3 -c 'import sys;print(sys.executable)')
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
for nw in 17 100 255; do
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1_$nw -- $PY $R/tools/bench_lines.py 1000000 $nw 3 > $OUT/sq1_$nw.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/sq2_$nw -- $PY $R/tools/bench_lines.py 1000000 $nw 3 > $OUT/sq2_$nw.log 2>&1
  python3 $R/tools/summarize_pmc.py $OUT/sq1_$nw stream_lines > $OUT/sq1_$nw.json
  python3 $R/tools/summarize_pmc.py $OUT/sq2_$nw stream_lines > $OUT/sq2_$nw.json
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $OUT/sq1_*.json $OUT/sq2_*.json
