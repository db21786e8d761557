#!/bin/bash
cd "$(dirname "$0")/../.."
for mb in 16 48 128 256; do
  for rep in 1 2; do
    GORTT_VERBOSE=1 GORTT_CHUNK_MB=$mb gort_amd/bin/gortt -LAI 4.0 --binary-in --binary-out < /tmp/gort_bin_in.dat > /dev/null
  done
done
