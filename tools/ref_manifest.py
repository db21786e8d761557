#!/usr/bin/env python3
"""Write oracle/_ref.MANIFEST: what the reference build under oracle/_ref/ (git-ignored; it travels to the GPU box with
the snapshot) IS - sha256 of every artefact bench.py and the tests load or run, sha256 of the reference sources it was
compiled from (hashes only, no text), compilers and flags.  bench.py prints the first 16 hex digits of
libgortt_ref.so's hash as `reference_build`, so that a BENCH record names the reference build it timed and checked
against; 'absent' there means the build did not travel and the CPU baseline fell back to the port.
Run by `make -C oracle ref` (build container only: needs /root/reference)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GORT_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def first_line(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, timeout=60).stdout.decode(errors="replace").splitlines()[0].strip()
    except Exception as ex:                                   # noqa: BLE001
        return "unavailable (%s)" % ex


def main():
    arts = ["gortt", "gortt_fp", "libgortt_ref.so"]
    missing = [a for a in arts if not os.path.exists(os.path.join(OUT, a))]
    if missing:
        sys.exit("ref_manifest: %s missing under oracle/_ref (run `make -C oracle ref`)" % missing)
    srcs = ["gortt.c", "gortt_pn_kopen.c", "gortt_brdf.c", "gortt_albedo.c", "gortt_lidar.c", "include/gortt.h",
            "include/soil_rho.h", "PROSPECT-D/dataSpec_PDB.f90", "PROSPECT-D/tav_abs.f90", "PROSPECT-D/prospect_DB.f90"]
    man = {
        "what": "the reference tquaife/gort compiled in place from %s by oracle/Makefile (own recipe; gcc for the C, AMD flang for "
                "the vendored PROSPECT-D Fortran); outputs only under oracle/_ref/" % REF,
        "artefacts_sha256": {a: sha(os.path.join(OUT, a)) for a in arts},
        "reference_build": sha(os.path.join(OUT, "libgortt_ref.so"))[:16],
        "sources_sha256": {s: sha(os.path.join(REF, s)) for s in srcs if os.path.exists(os.path.join(REF, s))},
        "shims_sha256": {s: sha(os.path.join(ROOT, "oracle", s)) for s in ("ref_shim.c", "ref_capture.c", "Makefile")},
        "compilers": {"cc": first_line(["gcc", "--version"]), "flang": first_line(["/opt/rocm/lib/llvm/bin/flang", "--version"])},
        "flags": {"c": "-O3 -g -w (the reference ships -Wall -g with -O3 commented out, makefile:3; output byte-identical)",
                  "fortran": "-O2 -w", "link": "-lm -static-libflang"},
    }
    path = os.path.join(ROOT, "oracle", "_ref.MANIFEST")
    with open(path, "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
        f.write("\n")
    print("ref_manifest: reference_build", man["reference_build"], "->", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main()
