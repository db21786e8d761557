"""Host-side logic of bench.py that decides or labels something (no GPU): the rank correlation of `per_draw`, the spread that
retires or keeps the placement selection, the environment of a single-process child under a launcher."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _draws(probes, kernels):
    return [{"probe_gbs": p, "kernel_ms": k} for p, k in zip(probes, kernels)]


def test_per_draw_rank_correlation_is_null_within_the_noise_and_a_number_beyond():
    """records of rounds 5 and 6.  Draws within 2 % of each other by the kernel's clock: nothing to order."""
    assert bench.probe_kernel_rank_correlation(_draws([6980, 6788, 7342], [6.95, 6.98, 6.96])) is None      # 8 % probe spread, 0.4 % kernel spread
    assert bench.probe_kernel_rank_correlation(_draws([7215, 7205, 7224], [6.819, 6.811, 6.802])) is None
    assert bench.probe_kernel_rank_correlation(None) is None and bench.probe_kernel_rank_correlation(_draws([7000], [7.0])) is None
    # one slow placement, found by the probe: rho = 1; a probe that ranks the draws backwards: -1; ties get half ranks
    assert bench.probe_kernel_rank_correlation(_draws([6270, 6980, 7114], [7.57, 7.00, 6.98])) == pytest.approx(1.0)
    assert bench.probe_kernel_rank_correlation(_draws([7278, 7188, 6321], [6.769, 6.783, 7.462])) == pytest.approx(1.0)
    assert bench.probe_kernel_rank_correlation(_draws([6000, 6500, 7000], [6.5, 7.0, 7.5])) == pytest.approx(-1.0)
    assert bench.probe_kernel_rank_correlation(_draws([7178, 7203, 7214], [7.02, 6.87, 6.93])) == pytest.approx(0.5)
    rho = bench.probe_kernel_rank_correlation(_draws([7000, 7000, 6000, 6500], [6.8, 6.8, 7.5, 7.2]))
    assert rho == pytest.approx(1.0)
    assert bench.probe_kernel_rank_correlation(_draws([7000, 7000, 7000], [6.5, 7.0, 7.5])) is None         # a constant probe orders nothing


def test_kernel_spread():
    assert bench.kernel_spread(None) is None and bench.kernel_spread([]) is None and bench.kernel_spread(_draws([1], [7.0])) is None
    assert bench.kernel_spread(_draws([1, 2, 3], [6.769, 6.783, 7.462])) == pytest.approx((7.462 - 6.769) / 6.769)


def test_single_process_env_drops_the_launchers_rendezvous(monkeypatch):
    """The traffic pass of an N > 1 run starts bench.py as a child of rank 0: it must come up as a world of one on that GPU."""
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "8"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500"),
                 ("TORCHELASTIC_RUN_ID", "x"), ("GROUP_RANK", "0"), ("LOCAL_WORLD_SIZE", "8"), ("HSA_ENABLE_IPC_MODE_LEGACY", "0")):
        monkeypatch.setenv(k, v)
    env = bench._single_process_env()
    assert not [k for k in env if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")
                or k.startswith("TORCHELASTIC_")]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "PATH" in env
