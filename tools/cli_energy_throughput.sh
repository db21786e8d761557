#!/bin/bash
# End-to-end time of `gortt -energy --binary-in --binary-out` (process start, GPU init and gap probabilities included):
# N random lines x 2101 bands, 91 distinct sun zeniths, output (rsurf + albedo, favegt, fasoil per band: 67 KB per line)
# to /dev/null; with the rows of equal sun directions shared (default) and with every line evaluated
# (GORT_ENERGY_DEDUP=0).   usage: tools/cli_energy_throughput.sh [lines]
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
for dedup in 1 0 1 0; do
GORT_ENERGY_DEDUP=$dedup GORTT_VERBOSE=1 python3 - "$N" "$dedup" <<'PY'
import subprocess, sys, time
n, dedup = int(sys.argv[1]), sys.argv[2]
t0 = time.perf_counter()
r = subprocess.run(["gort_amd/bin/gortt", "-LAI", "4.0", "-energy", "--binary-in", "--binary-out"], stdin=open("/tmp/gort_bin_in.dat", "rb"),
                   stdout=open("/dev/null", "wb"), stderr=subprocess.PIPE, check=True)
dt = time.perf_counter() - t0
print("gortt -energy --binary-in --binary-out, %d lines x 2101 bands -> /dev/null, GORT_ENERGY_DEDUP=%s: %.3f s  (%.1f GB/s of rows)"
      % (n, dedup, dt, n * (4 + 4 * 2101) * 8 / dt / 1e9))
print("   " + r.stderr.decode().strip().replace("\n", "\n   "))
PY
done
