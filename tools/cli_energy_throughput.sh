#!/bin/bash
# End-to-end time of `gortt -energy` (process start, GPU init and gap probabilities included), N random lines with 91
# distinct sun zeniths, output to /dev/null:
#   --binary-in --binary-out, 2101 bands (rsurf + albedo, favegt, fasoil per band: 67 KB per line)
#   text in and out, 180 bands (what the reference's header can hold)
# each with the albedo rows in the indexed form (default: every distinct row evaluated, copied and formatted once per chunk)
# and with a row per line (GORTT_ENERGY_DENSE=1, the path of rounds 3-4).   usage: tools/cli_energy_throughput.sh [lines]
set -e
N=${1:-1000000}
cd "$(dirname "$0")/.."
python3 - "$N" <<'PY'
import sys, numpy as np
n = int(sys.argv[1])
rng = np.random.default_rng(1)
a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 90, n).astype(float), np.zeros(n)], 1)
wl = np.arange(400, 2501)
with open("/tmp/gort_bin_in.dat", "wb") as f:
    f.write(("%d %d %s\n" % (n, len(wl), " ".join(map(str, wl)))).encode())
    f.write(a.astype("<f8").tobytes())
wl = np.linspace(400, 2500, 180).round(1)
with open("/tmp/gort_txt_in.dat", "wb") as f:
    f.write(("%d %d %s\n" % (n, len(wl), " ".join("%g" % w for w in wl))).encode())
    np.savetxt(f, a, fmt="%.4f")
PY
for mode in binary text; do
for dense in 0 1 0 1; do
GORTT_ENERGY_DENSE=$dense GORTT_VERBOSE=1 python3 - "$N" "$dense" "$mode" <<'PY'
import subprocess, sys, time
n, dense, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
args = ["--binary-in", "--binary-out"] if mode == "binary" else []
src = "/tmp/gort_bin_in.dat" if mode == "binary" else "/tmp/gort_txt_in.dat"
nw = 2101 if mode == "binary" else 180
t0 = time.perf_counter()
r = subprocess.run(["gort_amd/bin/gortt", "-LAI", "4.0", "-energy"] + args, stdin=open(src, "rb"),
                   stdout=open("/dev/null", "wb"), stderr=subprocess.PIPE, check=True)
dt = time.perf_counter() - t0
print("gortt -energy %s, %d lines x %d bands -> /dev/null, GORTT_ENERGY_DENSE=%s: %.3f s  (%.2e rows of 4 x nw numbers per s)"
      % (" ".join(args) or "(text)", n, nw, dense, dt, n / dt))
print("   " + r.stderr.decode().strip().replace("\n", "\n   "))
PY
done
done
