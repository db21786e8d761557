#!/bin/bash
# What more independent chains per SIMD buy the angle-only stage (two lines per lane would double the chains of
# stream_lines_kernel's geometry phase, which runs at 2.25 waves per SIMD): geometry_stream_kernel<fused> - a thread per line, the same
# geometry_core, 112 VGPRs, four waves per SIMD - held to 1 / 2 / 3 waves per SIMD by unused LDS (probe builds:
# -DGORT_PROBE_GEOM_LDS_PAD=120000 / 70000 / 50000 in tools/probes/ab_libs/geomocc{1,2,3}.so), a million lines x 1 band.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for lib in tools/probes/ab_libs/geomocc1.so tools/probes/ab_libs/geomocc2.so tools/probes/ab_libs/geomocc3.so gort_amd/libgort_amd.so; do
    echo -n "$(basename $lib .so): "; GORT_AMD_LIB=$PWD/$lib python3 tools/bench_lines.py 1000000 1 15 2>/dev/null | grep lines
  done
done
