"""AddressSanitizer + UBSan over the CPU-side code: the oracle and the product's host translation unit
(canopy derivation, spectra, LUT text, formatter).  GPU sanitizers are not available on this pool."""
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
