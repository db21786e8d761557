"""AddressSanitizer + UBSan over the CPU-side code: the oracle, the product's host translation unit (canopy derivation,
spectra, LUT text, formatter) and the front end of the `gortt` executable (command line, header, the scanf emulation of
the angle lines, row formatter: gort_amd/csrc/gortt_cli.h) on every reference-generated CLI case.  GPU sanitizers are
not available on this pool."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
DATA = '-DGORT_DATA_DIR="%s"' % os.path.join(ROOT, "gort_amd", "data")
DRIVER = os.path.join(ROOT, "tests", "sanitize_driver.c")


def _run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, timeout=600, **kw)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc missing")
def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    b = _run(["gcc", "-std=gnu99", "-ffp-contract=off", "-DDRIVE_ORACLE", DATA, "-I" + os.path.join(ROOT, "oracle")] + SAN +
             [DRIVER, os.path.join(ROOT, "oracle", "gort_oracle.c"), "-o", exe, "-lm"])
    assert b.returncode == 0, b.stderr.decode()
    r = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"oracle sanitize driver ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_product_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_san")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gort_amd", "csrc")]
    obj = str(tmp_path / "driver.o")
    b = _run(["gcc", "-std=gnu99", "-c"] + inc + SAN + [DRIVER, "-o", obj])
    assert b.returncode == 0, b.stderr.decode()
    b = _run(["g++", "-std=c++17", "-ffp-contract=off", DATA] + inc + SAN +
             [os.path.join(ROOT, "gort_amd", "csrc", "gort_host.cpp"), obj, "-o", exe, "-lm"])
    assert b.returncode == 0, b.stderr.decode()
    r = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"host sanitize driver ok" in r.stdout


# messages the executable can end with before any line reaches the device (gortt.c:153-184, 232-237, 1003-1136, 1286-1328)
_FRONT_END = ("unknown option on command line", "unknown argument on command line", "needs a value", "error reading data on stdin",
              "error reading number of", "expected number of wavelengths", "expected number of angles", "error on input, line",
              "wavlength out of range", "error opening probability file", "-soil_spectra is not supported", "--gpus needs")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_cli_front_end_under_asan_ubsan(tmp_path):
    """Every CLI case the reference generated (50 hand-picked + 160 random command lines + 100 hostile ones + 269 oddly
    spelled numbers + a 4000-line stream) through the executable's front end, built with the sanitizers, on the CPU: no report, the exit code
    and stderr of the reference wherever the run ends in front of the device, and the echo of the four angles of every
    line (what scanf("%lf %lf %lf %lf") made of it) equal to the first four numbers of the reference's row."""
    exe = str(tmp_path / "gortt_dry")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gort_amd", "csrc")]
    b = _run(["g++", "-std=c++17", "-ffp-contract=off", "-pthread", DATA] + inc + SAN +
             [os.path.join(ROOT, "tests", "support", "cli_dry_run.cpp"), os.path.join(ROOT, "gort_amd", "csrc", "gort_host.cpp"),
              "-o", exe, "-lm"])
    assert b.returncode == 0, b.stderr.decode()[-3000:]
    golden = os.path.join(ROOT, "tests", "golden")
    cases = []
    for f in ("cli_cases.json", "cli_fuzz_cases.json", "cli_hostile_cases.json", "cli_number_format_cases.json",
              "cli_scanf_corner_cases.json"):
        cases += json.load(open(os.path.join(golden, f), encoding="utf-8"))
    import gzip
    bulk = json.load(gzip.open(os.path.join(golden, "cli_bulk.json.gz"), "rt", encoding="utf-8"))      # one 4000-line stream: chunked reads
    cases.append(dict(bulk, name="bulk", rc=0))
    assert len(cases) >= 570
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", GORTT_THREADS="3")
    compared_rows = compared_errors = 0
    for case in cases:
        args = list(case["args"])
        if "@LUT@" in args:
            p = tmp_path / "lut.dat"
            p.write_text(case["lut_text"])
            args[args.index("@LUT@")] = str(p)
        run = subprocess.run(["gortt"] + args, executable=exe, input=case["stdin"].encode("utf-8"), capture_output=True, timeout=120, env=env)
        err = run.stderr.decode("latin-1")
        assert "Sanitizer" not in err and "runtime error" not in err, (case["name"], err[-2000:])
        assert run.returncode in (0, 1), (case["name"], run.returncode, err[-500:])
        if "-W" in args or "-u" in args or any(a.startswith("-W") or a.lower().startswith("-u") for a in args):
            continue                                            # the table is the device's; the usage text is compared on the GPU box
        if "invalid crown geometry" in err:
            continue                                            # documented deviation: refused where the reference crashes, loops or goes on
        ref_err = case["stderr"]
        front = any(m in ref_err for m in _FRONT_END)
        if case["rc"] != 0 and not front:
            continue                                            # ends behind the front end (a degenerate crown, ...): GPU tests
        assert run.returncode == case["rc"], (case["name"], args, err)
        assert err == ref_err, (case["name"], err, ref_err)
        compared_errors += case["rc"] != 0
        got, want = run.stdout.decode("latin-1").split("\n"), case["stdout"].split("\n")
        if want and want[0] and got[0] != want[0]:
            # no header echoed: the run ended before it (e.g. a wavelength out of range) in both programs
            assert got == [""] and case["rc"] != 0, (case["name"], got[:2], want[:2])
            continue
        assert len(got) == len(want), (case["name"], len(got), len(want))
        for g, w in zip(got[1:], want[1:]):
            assert g.split()[:4] == w.split()[:4], (case["name"], g, w)
            compared_rows += 1
    assert compared_rows > 5000 and compared_errors > 200, (compared_rows, compared_errors)
