#!/bin/bash
# The A/B suite N times in one box, one process each (a SIGSEGV of the measuring build was seen once in round 5):
# gpurun -- bash tools/ab_suite_repeat.sh [N] -> gpurun_out/ab_repeat.log (+ the faulthandler output of any run that died)
N=${1:-10}
mkdir -p gpurun_out
: > gpurun_out/ab_repeat.log
bad=0
for i in $(seq 1 $N); do
  GORT_AB_SUITE=1 GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_ab.so python -X faulthandler -m pytest tests -q -x -m "gpu and ab" -p no:cacheprovider > gpurun_out/ab_run_$i.log 2>&1
  rc=$?
  echo "run $i: rc $rc: $(tail -1 gpurun_out/ab_run_$i.log)" | tee -a gpurun_out/ab_repeat.log
  if [ $rc -eq 0 ]; then rm -f gpurun_out/ab_run_$i.log; else bad=$((bad + 1)); fi
done
echo "$bad of $N runs failed" | tee -a gpurun_out/ab_repeat.log
exit $bad
