"""Build libgort_amd.so (HIP kernels + C ABI, gfx950) and the `gortt` drop-in executable.

    python -m gort_amd.build [--force]
    python -m gort_amd.build --ab            (the measuring build with the A/B switches, gort_amd/libgort_amd_ab.so)
    python -m gort_amd.build --stamps        (the same with phase stamps in the kernels, gort_amd/libgort_amd_stamps.so)

hipcc cross-compiles for gfx950 without a GPU.  Everything is built in-tree:
gort_amd/libgort_amd.so and gort_amd/bin/gortt travel to the GPU box with the
repository snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "build")
LIB = os.path.join(PKG, "libgort_amd.so")
BIN = os.path.join(PKG, "bin", "gortt")
ARCH = "gfx950"

# (source, extra flags).  The gap kernel and the host precompute keep IEEE operation order
# (-ffp-contract=off): they feed integer truncations / are compared to the last bits.
UNITS = [
    ("gort_gap.hip", ["-ffp-contract=off"]),
    ("gort_geometry.hip", []),
    ("gort_tables.hip", []),
    ("gort_lut_expand.hip", []),
    ("gort_stream_expand.hip", []),
    ("gort_stream_lines.hip", []),
    ("gort_energy.hip", []),
    ("gort_xcd.hip", []),
    ("gort_pipe.hip", []),
    ("gort_spectra.hip", []),
    ("gort_api.hip", []),
    ("gort_rccl.cpp", []),
    ("gort_host.cpp", ["-ffp-contract=off", '-DGORT_DATA_DIR="%s"' % os.path.join(PKG, "data")]),
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("[gort_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _run_all(cmds, jobs=None):
    """Independent compile commands, a few at a time (hipcc is one process per unit; the container has 8 cores and no swap)."""
    if not cmds:
        return
    from concurrent.futures import ThreadPoolExecutor
    jobs = jobs or max(1, min(4, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(jobs) as pool:
        list(pool.map(_run, cmds))


def build_variant(name, defines, force=True):
    """gort_amd/libgort_amd_<name>.so: the same sources with extra -D flags - a measuring build beside the product library,
    never loaded unless GORT_AMD_LIB names it (or the `ab` tests do).
        ab      -DGORT_AB: the A/B environment switches and the alternative kernel instantiations they select (csrc/gort_internal.h);
                the tests marked `ab` run on it and hold every variant to the bits of the default
        stamps  -DGORT_STAMPS -DGORT_AB: phase stamps in the kernels as well (csrc/gort_stamps.h; tools/stamps.py)"""
    obj = os.path.join(SRC, "build_" + name)
    os.makedirs(obj, exist_ok=True)
    cc = hipcc()
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(SRC, "*.h"))) + [os.path.abspath(__file__)]
    common = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + ["-D" + d for d in defines] + \
             ["-I" + os.path.join(ROOT, "include"), "-I" + SRC, "--offload-arch=" + ARCH]
    objs, todo = [], []
    for src, extra in UNITS:
        s = os.path.join(SRC, src)
        o = os.path.join(obj, os.path.splitext(src)[0] + ".o")
        if force or _newer(o, [s] + headers):
            todo.append([cc] + common + extra + ["-c", s, "-o", o])
        objs.append(o)
    _run_all(todo)
    lib = os.path.join(PKG, "libgort_amd_%s.so" % name)
    if force or _newer(lib, objs):
        _run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs + ["-ldl"])
    return lib


def build_stamps():
    return build_variant("stamps", ["GORT_STAMPS", "GORT_AB"])


def build_ab(force=False):
    return build_variant("ab", ["GORT_AB"], force=force)


def build(force=False, verbose_resources=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    cc = hipcc()
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(SRC, "*.h"))) + [os.path.abspath(__file__)]
    data = [os.path.join(PKG, "data", f) for f in ("prospect_d_coeffs.f32", "price_soil_eofs.f64")]
    common = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
              "-I" + SRC, "--offload-arch=" + ARCH]
    if verbose_resources:
        common.append("-Rpass-analysis=kernel-resource-usage")
    objs, todo = [], []
    for src, extra in UNITS:
        s = os.path.join(SRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        deps = [s] + headers + (data if src == "gort_host.cpp" else [])
        if force or _newer(o, deps):
            todo.append([cc] + common + extra + ["-c", s, "-o", o])
        objs.append(o)
    _run_all(todo)
    if force or _newer(LIB, objs):
        _run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"])
    main = os.path.join(SRC, "gortt_main.cpp")
    if os.path.exists(main) and (force or _newer(BIN, [main, LIB] + headers)):
        _run([cc, "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), main, "-o", BIN,
              "-L" + PKG, "-lgort_amd", "-pthread", "-Wl,-rpath,$ORIGIN/.."])
    return LIB


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build_stamps()
    elif "--ab" in sys.argv:
        build_ab(force="--force" in sys.argv)
    else:
        build(force="--force" in sys.argv, verbose_resources="--resources" in sys.argv)
