#!/bin/bash
cd "$(dirname "$0")/../.."
N=${1:-1000000}; M=${2:-180}
python3 - "$N" "$M" > /tmp/gort_stream.txt <<'PY'
import sys, numpy as np
n, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
wl = np.linspace(400, 2500, m).round().astype(int)
print(n, m, " ".join(map(str, wl)))
a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
sys.stdout.write("".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in a))
PY
nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
for t in "" 8 16 32 64; do
  echo "GORTT_THREADS=$t"; GORTT_VERBOSE=1 GORTT_THREADS=$t gort_amd/bin/gortt -LAI 4.0 < /tmp/gort_stream.txt > /dev/null
done
