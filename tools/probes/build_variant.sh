#!/bin/bash
# build a variant library with extra -D flags: tools/probes/build_variant.sh <suffix> <flags...>  -> gort_amd/libgort_amd_<suffix>.so
set -e
cd "$(dirname "$0")/../.."
SUF=$1; shift
OBJ=/tmp/gort_variant_$SUF; mkdir -p $OBJ
COMMON="-O3 -fPIC -std=c++17 -Iinclude -Igort_amd/csrc --offload-arch=gfx950 $*"
hipcc $COMMON -ffp-contract=off -c gort_amd/csrc/gort_gap.hip -o $OBJ/gap.o &
hipcc $COMMON -c gort_amd/csrc/gort_brdf.hip -o $OBJ/brdf.o &
hipcc $COMMON -c gort_amd/csrc/gort_stream.hip -o $OBJ/stream.o &
hipcc $COMMON -c gort_amd/csrc/gort_pipe.hip -o $OBJ/pipe.o &
hipcc $COMMON -c gort_amd/csrc/gort_spectra.hip -o $OBJ/spectra.o &
hipcc $COMMON -c gort_amd/csrc/gort_api.hip -o $OBJ/api.o &
hipcc $COMMON -ffp-contract=off -DGORT_DATA_DIR="\"$PWD/gort_amd/data\"" -c gort_amd/csrc/gort_host.cpp -o $OBJ/host.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o gort_amd/libgort_amd_$SUF.so $OBJ/*.o
ls -la gort_amd/libgort_amd_$SUF.so
