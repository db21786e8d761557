#!/bin/bash
# The stream family's sample in 22 + 1 instructions (round 6) against the 24 + 1 of rounds 3-5, on ONE box:
# gort_amd/libgort_amd_prev.so is the tree before the change (built by hand from the parent commit), libgort_amd.so the tree as
# it is.  Alternating, so that a drift of the box shows.   tools/probes/sample_form_ab.sh > gpurun_out/sample_form_ab.log
cd "$(dirname "$0")/../.."
PREV=$PWD/gort_amd/libgort_amd_prev.so
NEW=$PWD/gort_amd/libgort_amd.so
for round in 1 2; do
  for lib in "$PREV" "$NEW"; do
    echo "== $(basename $lib) (pass $round)"
    GORT_AMD_LIB=$lib python3 tools/bench_lines.py 1000000 100 15 || exit 1
    GORT_AMD_LIB=$lib python3 tools/bench_lines.py 1000000 32 15 || exit 1
    GORT_AMD_LIB=$lib python3 tools/bench_lines.py 1000000 190 15 || exit 1
    GORT_AMD_LIB=$lib python3 tools/bench_lines.py 65536 2101 15 || exit 1
    GORT_AMD_LIB=$lib python3 tools/bench_lines.py 1048576 2101 9 || exit 1
    GORT_AMD_LIB=$lib python3 tools/probes/members_stream.py 1000 1000 100 640 2101 || exit 1
  done
done
