#!/bin/bash
# Where the time of the drop-in `gortt` goes for a long stream: text/binary on either side.
# usage: tools/cli_breakdown.sh [lines] [bands]
set -e
N=${1:-1000000}; M=${2:-180}
cd "$(dirname "$0")/../.."
python3 - "$N" "$M" <<'PY'
import sys, numpy as np
n, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
wl = np.linspace(400, 2500, m).round().astype(int)
head = "%d %d %s\n" % (n, m, " ".join(map(str, wl)))
a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
open("/tmp/gs.txt", "w").write(head + "".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in a))
open("/tmp/gs.bin", "wb").write(head.encode() + a.tobytes())
PY
t() { local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$(( (t1 - t0) / 1000000 )) ms"; }
echo "text in, text out   : $(t sh -c 'gort_amd/bin/gortt -LAI 4.0 < /tmp/gs.txt > /tmp/go.txt')"
echo "text in, binary out : $(t sh -c 'gort_amd/bin/gortt -LAI 4.0 --binary-out < /tmp/gs.txt > /tmp/go.bin')"
echo "binary in, binary out: $(t sh -c 'gort_amd/bin/gortt -LAI 4.0 --binary-in --binary-out < /tmp/gs.bin > /tmp/go2.bin')"
echo "binary in, text out : $(t sh -c 'gort_amd/bin/gortt -LAI 4.0 --binary-in < /tmp/gs.bin > /tmp/go2.txt')"
echo "text out to /dev/null: $(t sh -c 'gort_amd/bin/gortt -LAI 4.0 < /tmp/gs.txt > /dev/null')"
echo "1 line (start-up)   : $(t sh -c 'head -2 /tmp/gs.txt | sed "1s/^[0-9]* /1 /" | gort_amd/bin/gortt -LAI 4.0 > /dev/null')"
cmp /tmp/go.txt /tmp/go2.txt && echo "text outputs identical"; cmp /tmp/go.bin /tmp/go2.bin && echo "binary outputs identical"
