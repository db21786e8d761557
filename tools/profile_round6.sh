#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun; outputs under gpurun_out/prof6/, the summaries are then committed
# under profiles/r06/prof/).  The interpreter binary itself follows `--` (no env / bash / launcher hop under rocprofv3).
#   A. headline bench: unprofiled (it measures its own HBM traffic and the configs' instruction counts with --pmc child
#      passes); kernel-trace + stats of the same steps, and the timed launches alone (warm-ups dropped) from the trace
#   B. the line kernel: shape scan, kernel stats and an SQ pass at 100 bands
#   C. `-energy`: the dense stream forms (measuring build), the CLI end to end (indexed against dense)
#   D. the ALU / latency bound configs (C2, C3, C4): times, stats, SQ pass, phase stamps
#   E. member grids below 128 bands: 1000 members x 7 bands (and x 100)
#   F. the few-band LUT forms across band counts and under the HBM counters; the `-energy` table pass by number of sun directions
#   G. [r6] the ensemble observation operator (members stream): times across band counts, kernel trace, traffic and SQ counters
#   H. [r6] the wide stream (flat panels) and a hemisphere x 100 / 127 / 128 bands per kernel
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof6
PY=$(python3 -c 'import sys;print(sys.executable)')
mkdir -p "$OUT"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
prof() { # name, then rocprofv3 args..., then -- program
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 "$@" ) > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
which=${1:-ABCDEFGH}
if [[ $which == *A* ]]; then
  cd "$R" && timeout -k 10 700 $PY bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"; echo "bench rc=$?"
  prof bench_stats --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- $PY "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-parity --sustain-s 0 --no-config5 --no-configs --no-traffic --placement-evidence 0
  for f in "$OUT"/bench_stats/*/*_kernel_trace.csv; do
    [ -f "$f" ] && $PY "$R/tools/timed_launches.py" "$f" expand_flat_kernel 3 20 > "$OUT/bench_kernel_timed_only.json"
  done
fi
if [[ $which == *B* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/shape_scan.py > "$OUT/shape_scan.log" 2>&1
  prof lines_100_stats --kernel-trace --stats --output-format csv -d "$OUT/lines_100_stats" -- $PY "$R/tools/bench_lines.py" 1000000 100 20
  prof stream_100_sq --pmc $SQ --output-format csv -d "$OUT/stream_100_sq" -- $PY "$R/tools/bench_lines.py" 1000000 100 3
  prof lines_100_pmc_write --pmc WRITE_SIZE --output-format csv -d "$OUT/lines_100_pmc_write" -- $PY "$R/tools/bench_lines.py" 1000000 100 3
  prof lines_100_pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/lines_100_pmc_fetch" -- $PY "$R/tools/bench_lines.py" 1000000 100 3
fi
if [[ $which == *C* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/bench_energy_stream.py 1048576 2101 > "$OUT/energy_stream.log" 2>&1
  cd "$R" && timeout -k 10 400 tools/cli_energy_throughput.sh 1000000 > "$OUT/cli_energy_throughput.log" 2>&1
fi
if [[ $which == *D* ]]; then
  cd "$R" && timeout -k 10 200 $PY tools/bench_configs.py > "$OUT/configs.log" 2>&1
  prof configs_stats --kernel-trace --stats --output-format csv -d "$OUT/configs_stats" -- $PY "$R/tools/bench_configs.py"
  prof configs_sq --pmc $SQ --output-format csv -d "$OUT/configs_sq" -- $PY "$R/tools/bench_configs.py"
  cd "$R" && timeout -k 10 300 $PY tools/stamps.py c3 c4 > "$OUT/stamps.log" 2>&1
fi
if [[ $which == *E* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/bench_ensemble.py 1000 1000 7 > "$OUT/ensemble_few_bands.log" 2>&1
  cd "$R" && timeout -k 10 300 $PY tools/bench_ensemble.py 1000 250 100 >> "$OUT/ensemble_few_bands.log" 2>&1
fi
if [[ $which == *F* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/probes/mid_band_grid.py > "$OUT/mid_band_grid.log" 2>&1
  cd "$R" && timeout -k 10 300 tools/probes/few_band_lut_pmc.sh > "$OUT/few_band_lut_pmc.log" 2>&1
  cd "$R" && timeout -k 10 300 $PY tools/probes/energy_table_cost.py > "$OUT/energy_table_cost.log" 2>&1
fi
if [[ $which == *G* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/probes/members_stream.py 1000 1000 > "$OUT/members_stream.log" 2>&1
  cd "$R" && timeout -k 10 900 bash tools/prof_members.sh gort_amd/libgort_amd.so > "$OUT/members_stream_counters.log" 2>&1
fi
if [[ $which == *H* ]]; then
  cd "$R" && { timeout -k 10 200 $PY tools/bench_stream.py 1048576 10 "all distinct"; timeout -k 10 200 $PY tools/bench_stream.py 65536 20 "all distinct"; } > "$OUT/wide_stream.log" 2>&1
  cd "$R" && for nw in 100 127 128; do bash tools/probes/kernel_times.sh lut_$nw tools/probes/mid_band_grid.py $nw | head -4; done > "$OUT/lut_mid_bands_kernels.log" 2>&1
fi
# ---- summaries
cd "$R"
for t in bench_stats lines_100_stats configs_stats; do
  for f in "$OUT/$t"/*/*_kernel_stats.csv; do [ -f "$f" ] && cp "$f" "$OUT/${t%_stats}_kernel_stats.csv"; done
done
for t in lines_100_pmc_write lines_100_pmc_fetch stream_100_sq configs_sq; do
  [ -d "$OUT/$t" ] && $PY tools/summarize_pmc.py "$OUT/$t" > "$OUT/$t.json" 2>> "$OUT/summarize.err"
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +     # raw per-dispatch CSVs: summarised above
du -sh "$OUT"; ls "$OUT"
