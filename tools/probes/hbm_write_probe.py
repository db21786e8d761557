#!/usr/bin/env python3
"""Measure what the box's HBM actually sustains for the access shapes this path uses:
write-only streams (the LUT kernel is one), read-only, and copy.  Context for the
`roofline.frac` in bench.py: the spec peak (8 TB/s) is quoted there, this probe says how
far a plain fill gets on the same box.  Run on the GPU box:  python tools/hbm_write_probe.py
"""
import sys
import time

import torch


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    for gb in (2, 8, 50):
        n = int(gb * 1e9 / 8)
        x = torch.empty(n, dtype=torch.float64, device="cuda")
        t = timed(lambda: x.zero_())
        print("zero_      %3d GB: %7.3f ms  %7.1f GB/s (write-only)" % (gb, t * 1e3, n * 8 / t / 1e9))
        t = timed(lambda: x.fill_(1.5))
        print("fill_      %3d GB: %7.3f ms  %7.1f GB/s (write-only)" % (gb, t * 1e3, n * 8 / t / 1e9))
        t = timed(lambda: x.sum())
        print("sum        %3d GB: %7.3f ms  %7.1f GB/s (read-only)" % (gb, t * 1e3, n * 8 / t / 1e9))
        y = torch.empty_like(x)
        t = timed(lambda: y.copy_(x))
        print("copy_      %3d GB: %7.3f ms  %7.1f GB/s (read+write bytes)" % (gb, t * 1e3, 2 * n * 8 / t / 1e9))
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.exit(main())
