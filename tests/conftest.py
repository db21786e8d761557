import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    config.addinivalue_line("markers", "ab: compares kernel forms through the A/B switches of the measuring build "
                                       "(gort_amd/libgort_amd_ab.so); runs in the process tests/test_ab_suite.py starts")


def pytest_collection_modifyitems(config, items):
    """The A/B switches do not exist in the product library: tests marked `ab` run in ONE process of their own on the
    measuring build (GORT_AB_SUITE=1, GORT_AMD_LIB=.../libgort_amd_ab.so: tests/test_ab_suite.py starts it) and are
    skipped everywhere else; that process runs nothing but them."""
    suite = os.environ.get("GORT_AB_SUITE") == "1"
    skip = pytest.mark.skip(reason="A/B test: runs on the measuring build, started by tests/test_ab_suite.py")
    keep, drop = [], []
    for item in items:
        is_ab = item.get_closest_marker("ab") is not None
        if suite and not is_ab:
            drop.append(item)
            continue
        if is_ab and not suite:
            item.add_marker(skip)
        keep.append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def relerr(a, b, floor=1e-300):
    """max |a-b|/max(|b|,floor) over finite reference entries; NaN pattern must match."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    m = np.isfinite(b)
    assert np.array_equal(np.isinf(a), np.isinf(b)), "inf pattern differs"
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), floor)))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
