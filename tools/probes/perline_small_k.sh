for n in 16384 65536 262144; do for rep in 1 2; do for k in 64 16 24 32; do for w in 2101 16808; do
echo -n "n=$n K=$k W=$w : "
GORT_STREAM_STEPS=$k GORT_STREAM_WAVES=$w timeout -k 10 100 python3 tools/bench_stream.py $n 15 "all" 2>&1 | grep "grouping=0" | cut -c42-100 || exit 1
done; done; done; done
