#!/bin/bash
# End-to-end rate of `gortt --binary-in --binary-out` (process start, GPU init and gap probabilities included):
# N random lines x 2101 bands, 91 distinct sun zeniths, output to /dev/null.   usage: tools/cli_binary_throughput.sh [lines]
set -e
N=${1:-1000000}
cd "$(dirname "$0")/.."
python3 - "$N" > /tmp/gort_bin_in.dat <<'PY'
import sys, numpy as np
n = int(sys.argv[1])
rng = np.random.default_rng(1)
wl = np.arange(400, 2501)
sys.stdout.buffer.write(("%d %d %s\n" % (n, len(wl), " ".join(map(str, wl)))).encode())
a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 90, n).astype(float), np.zeros(n)], 1)
sys.stdout.buffer.write(a.astype("<f8").tobytes())
PY
for rep in 1 2 3; do
python3 - "$N" <<'PY'
import subprocess, sys, time
n = int(sys.argv[1])
t0 = time.perf_counter()
subprocess.run(["gort_amd/bin/gortt", "-LAI", "4.0", "--binary-in", "--binary-out"], stdin=open("/tmp/gort_bin_in.dat", "rb"), stdout=open("/dev/null", "wb"), check=True)
dt = time.perf_counter() - t0
print("gortt --binary-in --binary-out, %d lines x 2101 bands -> /dev/null: %.3f s  %.3e samples/s  (%.1f GB/s of rows)" % (n, dt, n * 2101 / dt, n * 2105 * 8 / dt / 1e9))
PY
done
