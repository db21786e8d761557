for v in "" "GORT_EXPAND_NT=0" "GORT_EXPAND_XCD=0" "GORT_EXPAND_XCD=2" ""; do
echo -n "[$v] "; env $v timeout -k 10 100 python3 tools/bench_stream.py 1048576 10 "all" 2>&1 | grep "grouping=0" | cut -c42-100
done
