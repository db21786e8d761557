#!/bin/bash
# Round-4 profiles (run on the GPU box through gpurun; outputs under gpurun_out/prof4/, the summaries are then committed
# under profiles/r04/prof/).  The interpreter binary itself follows `--` (no env / bash / launcher hop under rocprofv3).
#   A. headline bench: unprofiled (it measures its own HBM traffic with two --pmc child passes), kernel-trace stats
#   B. the stream kernels: shape scan; the lines kernel at 100 bands: kernel stats, WRITE_SIZE / FETCH_SIZE passes;
#      SQ passes at 7 / 17 / 100 / 255 bands (instructions per wave, VALU-active share, waits)
#   C. `-energy` stream (1M lines x 2101 bands): shared rows, every line, every line its own sun; kernel stats
#   D. the ALU / latency bound configs (C2, C3, C4): stats + SQ pass
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof4
PY=$(python3 -c 'import sys;print(sys.executable)')
rm -rf "$OUT"; mkdir -p "$OUT"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
prof() { # name, then rocprofv3 args..., then -- program
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 "$@" ) > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
which=${1:-ABCD}
if [[ $which == *A* ]]; then
  cd "$R" && timeout -k 10 600 $PY bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"; echo "bench rc=$?"
  prof bench_stats --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- $PY "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-parity --sustain-s 0 --no-config5 --no-configs --no-traffic --lut-draws 1
fi
if [[ $which == *B* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/shape_scan.py > "$OUT/shape_scan.log" 2>&1
  prof lines_100_stats --kernel-trace --stats --output-format csv -d "$OUT/lines_100_stats" -- $PY "$R/tools/bench_lines.py" 1000000 100 20
  prof lines_100_pmc_write --pmc WRITE_SIZE --output-format csv -d "$OUT/lines_100_pmc_write" -- $PY "$R/tools/bench_lines.py" 1000000 100 3
  prof lines_100_pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/lines_100_pmc_fetch" -- $PY "$R/tools/bench_lines.py" 1000000 100 3
  for nw in 7 17 100 255; do
    prof stream_${nw}_sq --pmc $SQ --output-format csv -d "$OUT/stream_${nw}_sq" -- $PY "$R/tools/bench_lines.py" 1000000 $nw 3
  done
  for n in 65536 1048576; do
    cd "$R" && timeout -k 10 200 $PY tools/bench_stream.py $n 20 > "$OUT/stream_${n}_unprofiled.log" 2>&1
  done
fi
if [[ $which == *C* ]]; then
  cd "$R" && timeout -k 10 300 $PY tools/bench_energy_stream.py 1048576 2101 > "$OUT/energy_stream.log" 2>&1
  cd "$R" && timeout -k 10 300 $PY tools/bench_energy_stream.py 1048576 7 >> "$OUT/energy_stream.log" 2>&1
  prof energy_stats --kernel-trace --stats --output-format csv -d "$OUT/energy_stats" -- $PY "$R/tools/bench_energy_stream.py" 1048576 2101
fi
if [[ $which == *D* ]]; then
  cd "$R" && timeout -k 10 200 $PY tools/bench_configs.py > "$OUT/configs.log" 2>&1
  prof configs_stats --kernel-trace --stats --output-format csv -d "$OUT/configs_stats" -- $PY "$R/tools/bench_configs.py"
  prof configs_sq --pmc $SQ --output-format csv -d "$OUT/configs_sq" -- $PY "$R/tools/bench_configs.py"
fi
# ---- summaries
cd "$R"
for t in bench_stats lines_100_stats energy_stats configs_stats; do
  for f in "$OUT/$t"/*/*_kernel_stats.csv; do [ -f "$f" ] && cp "$f" "$OUT/${t%_stats}_kernel_stats.csv"; done
done
for t in lines_100_pmc_write lines_100_pmc_fetch stream_7_sq stream_17_sq stream_100_sq stream_255_sq configs_sq; do
  [ -d "$OUT/$t" ] && $PY tools/summarize_pmc.py "$OUT/$t" > "$OUT/$t.json" 2>> "$OUT/summarize.err"
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +     # raw per-dispatch CSVs: summarised above
du -sh "$OUT"; ls "$OUT"
