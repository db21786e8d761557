#!/bin/bash
# A/B of stream_lines_kernel builds on ONE box: each library in tools/probes/ab_libs/ (and the product) x band counts, interleaved twice
cd "$(dirname "$0")/../.."
PY=$(python3 -c "import sys;print(sys.executable)")
for rep in 1 2; do
for lib in gort_amd/libgort_amd.so tools/probes/ab_libs/*.so; do
  for nw in 32 100 127 190; do
    echo -n "$(basename $lib) " ; GORT_AMD_LIB=$PWD/$lib $PY tools/bench_lines.py 1000000 $nw 15 2>/dev/null | grep lines
  done
done
done
