#!/usr/bin/env python3
"""Where the ~0.3 s of a one-line `gortt` run go (the README example: 1 line x 4 bands; the reference takes ~0.2 s, all of
it its gap probabilities on one core).  Run on a GPU box from the repo root; profiles/r04/cli_startup.log."""
import os, subprocess, sys, time
R = os.getcwd()
IN = b"1 4 450 600 800 1000\n10 0 30 20\n"
GORTT = os.path.join(R, "gort_amd", "bin", "gortt")


def t(label, cmd, env=None, reps=7):
    e = dict(os.environ, **(env or {}))
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        subprocess.run(cmd, input=IN, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e)
        best = min(best, time.perf_counter() - t0)
    print("%-74s best of %d: %.3f s" % (label, reps, best), flush=True)


open("/tmp/hip_hello.cpp", "w").write("#include <hip/hip_runtime.h>\nint main() { void *p = nullptr; return hipMalloc(&p, 1 << 20) == hipSuccess ? 0 : 1; }\n")
subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "/tmp/hip_hello.cpp", "-o", "/tmp/hip_hello"], stderr=subprocess.DEVNULL)
t("a bare HIP process (hipMalloc of 1 MB and exit)", ["/tmp/hip_hello"])
t("gortt -LAI 4.0, README example", [GORTT, "-LAI", "4.0"])
t("  the same, GORTT_FAST_EXIT=1", [GORTT, "-LAI", "4.0"], {"GORTT_FAST_EXIT": "1"})
t("  + ROCR_VISIBLE_DEVICES=0 (one device visible)", [GORTT, "-LAI", "4.0"], {"GORTT_FAST_EXIT": "1", "ROCR_VISIBLE_DEVICES": "0"})
t("  + HIP_ENABLE_DEFERRED_LOADING=0 (every code object loaded at start)", [GORTT, "-LAI", "4.0"], {"GORTT_FAST_EXIT": "1", "HIP_ENABLE_DEFERRED_LOADING": "0"})
os.makedirs("/tmp/gortt_cache", exist_ok=True)
subprocess.run([GORTT, "-LAI", "4.0", "--lut-cache", "/tmp/gortt_cache"], input=IN, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t("  + gap tables from --lut-cache (no gap kernel, one kernel launch in all)", [GORTT, "-LAI", "4.0", "--lut-cache", "/tmp/gortt_cache"], {"GORTT_FAST_EXIT": "1"})
v = subprocess.run([GORTT, "-LAI", "4.0"], input=IN, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GORTT_VERBOSE="1"))
print("\n".join(l for l in v.stderr.decode().splitlines() if l.startswith("gortt:")))
ref = os.path.join(R, "oracle", "_ref", "gortt")
if os.access(ref, os.X_OK):
    t("the reference (oracle/_ref/gortt -LAI 4.0), one host core", [ref, "-LAI", "4.0"])
print("libgort_amd.so: %d bytes (all kernels: the LUT, placement and ensemble ones included)" % os.path.getsize(os.path.join(R, "gort_amd", "libgort_amd.so")))
