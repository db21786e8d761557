cd "$(dirname "$0")/../.."
export GORT_AMD_LIB=$PWD/gort_amd/libgort_amd_ab.so
for n in 65536 1048576; do
 for w in 2101 4202; do
  for k in 6 8 12 16 20 24; do
    echo -n "n=$n waves=$w steps=$k: "
    GORT_STREAM_WAVES=$w GORT_STREAM_STEPS=$k python3 tools/bench_stream.py $n 8 "all distinct" 2>/dev/null | tail -1
  done
 done
 echo -n "n=$n default: "; python3 tools/bench_stream.py $n 8 "all distinct" 2>/dev/null | tail -1
done
