#!/bin/bash
# End-to-end throughput of the drop-in `gortt` (text in, text out) vs the reference binary, same stream.
# usage: tools/cli_throughput.sh [lines] [bands]
set -e
N=${1:-200000}; M=${2:-180}
cd "$(dirname "$0")/.."
now() { python3 -c 'import time; print(time.time())'; }
rate() { python3 -c "import sys; dt=$2-$1; print('%.2f s, %.3e samples/s' % (dt, $3*$M/dt))"; }
python3 - "$N" "$M" > /tmp/gort_stream.txt <<'PY'
import sys, numpy as np
n, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
wl = np.linspace(400, 2500, m).round().astype(int)
print(n, m, " ".join(map(str, wl)))
a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
sys.stdout.write("".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in a))
PY
t0=$(now); gort_amd/bin/gortt -LAI 4.0 < /tmp/gort_stream.txt > /tmp/gort_out_gpu.txt; t1=$(now)
echo "gort_amd gortt, $N lines x $M bands (incl. GPU init + gap probabilities): $(rate $t0 $t1 $N)"
ls -la /tmp/gort_out_gpu.txt | awk '{print "output bytes:", $5}'
if [ -x oracle/_ref/gortt ]; then
  oracle/_ref/gortt -LAI 4.0 -W > /tmp/gort_lut.dat
  H=$(( N / 10 )); head -1 /tmp/gort_stream.txt | sed "s/^$N /$H /" > /tmp/gort_stream_small.txt; sed -n "2,$((H+1))p" /tmp/gort_stream.txt >> /tmp/gort_stream_small.txt
  t0=$(now); oracle/_ref/gortt -LAI 4.0 -P /tmp/gort_lut.dat < /tmp/gort_stream_small.txt > /tmp/gort_out_ref.txt; t1=$(now)
  echo "reference gortt -P, first $H lines: $(rate $t0 $t1 $H)"
  head -1 /tmp/gort_stream.txt | sed "s/^$N /2000 /" > /tmp/gort_s2.txt; sed -n "2,2001p" /tmp/gort_stream.txt >> /tmp/gort_s2.txt
  oracle/_ref/gortt -LAI 4.0 < /tmp/gort_s2.txt > /tmp/gort_r2.txt; gort_amd/bin/gortt -LAI 4.0 < /tmp/gort_s2.txt > /tmp/gort_g2.txt
  cmp /tmp/gort_r2.txt /tmp/gort_g2.txt && echo "2000-line x $M-band output: byte-identical to the reference (direct path, no LUT file)"
fi
