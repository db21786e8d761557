#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun; outputs under gpurun_out/prof3/, the summaries are then
# committed under profiles/r03/prof/).  The python program itself follows `--` (no env/bash hop under rocprofv3).
#   A. headline bench: unprofiled (it measures its own HBM traffic with two --pmc child passes), kernel-trace stats,
#      WRITE_SIZE / FETCH_SIZE passes of our own (cross-check), a run with that figure handed in
#   B. stream expansion at 65 536 and 1 048 576 random lines x 2101 bands (every line its own sun zenith): stats, HBM
#      passes, SQ pass
#   C. `-energy` stream (1M lines x 2101 bands, 91 sun zeniths): kernel trace with rows shared and with every line evaluated
#   D. the ALU / latency bound configs (C2, C3, C4): stats + SQ pass
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof3
COMMIT=${GORT_COMMIT:-unknown}
rm -rf "$OUT"; mkdir -p "$OUT"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
prof() { # name, then rocprofv3 args..., then -- program
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 "$@" ) > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
# ---- A
cd "$R" && timeout -k 10 500 python3 bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"; echo "bench rc=$?"
B="--steps 20 --warmup 3 --no-cpu-baseline --no-parity --sustain-s 0 --no-config5 --no-traffic"
prof bench_stats --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 "$R/bench.py" $B --lut-draws 1
prof bench_pmc_write --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --sustain-s 0 --no-config5 --no-traffic --lut-draws 1
prof bench_pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --sustain-s 0 --no-config5 --no-traffic --lut-draws 1
# ---- B
for n in 65536 1048576; do
  prof stream_${n}_stats --kernel-trace --stats --output-format csv -d "$OUT/stream_${n}_stats" -- python3 "$R/tools/bench_stream.py" $n 20 "all distinct"
  prof stream_${n}_pmc_write --pmc WRITE_SIZE --output-format csv -d "$OUT/stream_${n}_pmc_write" -- python3 "$R/tools/bench_stream.py" $n 3 "all distinct"
  prof stream_${n}_pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/stream_${n}_pmc_fetch" -- python3 "$R/tools/bench_stream.py" $n 3 "all distinct"
  prof stream_${n}_sq --pmc $SQ --output-format csv -d "$OUT/stream_${n}_sq" -- python3 "$R/tools/bench_stream.py" $n 3 "all distinct"
  cd "$R" && timeout -k 10 200 python3 tools/bench_stream.py $n 20 > "$OUT/stream_${n}_unprofiled.log" 2>&1
done
# ---- C
cd "$R" && timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 2101 > "$OUT/energy_stream.log" 2>&1
cd "$R" && timeout -k 10 300 python3 tools/bench_energy_stream.py 1048576 7 >> "$OUT/energy_stream.log" 2>&1
prof energy_stats --kernel-trace --stats --output-format csv -d "$OUT/energy_stats" -- python3 "$R/tools/bench_energy_stream.py" 1048576 2101
# ---- D
cd "$R" && timeout -k 10 200 python3 tools/bench_configs.py > "$OUT/configs.log" 2>&1
prof configs_stats --kernel-trace --stats --output-format csv -d "$OUT/configs_stats" -- python3 "$R/tools/bench_configs.py"
prof configs_sq --pmc $SQ --output-format csv -d "$OUT/configs_sq" -- python3 "$R/tools/bench_configs.py"
# ---- summaries
cd "$R"
for t in bench_stats stream_65536_stats stream_1048576_stats energy_stats configs_stats; do
  for f in "$OUT/$t"/*/*_kernel_stats.csv; do [ -f "$f" ] && cp "$f" "$OUT/${t%_stats}_kernel_stats.csv"; done
done
for t in bench_pmc_write bench_pmc_fetch stream_65536_pmc_write stream_65536_pmc_fetch stream_65536_sq stream_1048576_pmc_write stream_1048576_pmc_fetch stream_1048576_sq configs_sq; do
  python3 tools/summarize_pmc.py "$OUT/$t" > "$OUT/$t.json" 2>> "$OUT/summarize.err"
done
python3 - "$OUT" "$COMMIT" <<'PY'
import json, sys, os
out, commit = sys.argv[1], sys.argv[2]
w = json.load(open(out + "/bench_pmc_write.json")); f = json.load(open(out + "/bench_pmc_fetch.json"))
k = [n for n in w if "expand_flat_kernel" in n][0]
wkb, fkb = w[k]["WRITE_SIZE"], f[k]["FETCH_SIZE"]
traffic = (wkb + 2 * fkb) * 1024          # gfx950: FETCH_SIZE counts half the bytes of wide reads (MI355X_MICROARCH.md, HBM)
json.dump({"kernel": k, "workload": "91x91x361x2101", "n_gpus": 1, "WRITE_SIZE_KB": wkb, "FETCH_SIZE_KB": fkb, "commit": commit,
           "note": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in two separate passes (tools/profile_round3.sh, profiles/r03); "
                   "bytes = (WRITE_SIZE + 2*FETCH_SIZE)*1024: gfx950 FETCH_SIZE counts half the bytes of wide reads", "traffic_bytes": traffic},
          open(out + "/pmc_latest.json", "w"), indent=1)
print("traffic GB per launch:", traffic / 1e9)
open(out + "/traffic_gb.txt", "w").write("%.6f" % (traffic / 1e9))
PY
T=$(cat "$OUT/traffic_gb.txt" 2>/dev/null || echo "")
if [ -n "$T" ]; then
  cd "$R" && timeout -k 10 400 python3 bench.py --no-cpu-baseline --traffic-gb "$T" > "$OUT/bench_with_measured_traffic.json" 2>> "$OUT/bench_unprofiled.err"; echo "bench+traffic rc=$?"
fi
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +     # raw per-dispatch CSVs: summarised above
du -sh "$OUT"; ls "$OUT" | head -80
