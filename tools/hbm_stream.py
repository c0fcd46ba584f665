#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box (device fill, copy) -- the practical ceiling beside
the nominal 8 TB/s that the fill kernel's fraction is quoted against."""
import torch

n = 1 << 29                                  # 4 GiB of doubles
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: x.fill_(1.0))
print("fill   %6.2f GiB written : %7.0f GB/s (%.2f of 8 TB/s)" % (n * 8 / 2**30, n * 8 / t / 1e9, n * 8 / t / 8e12))
t = timed(lambda: x.zero_())
print("memset %6.2f GiB written : %7.0f GB/s (%.2f of 8 TB/s)" % (n * 8 / 2**30, n * 8 / t / 1e9, n * 8 / t / 8e12))
t = timed(lambda: y.copy_(x))
print("copy   %6.2f GiB moved   : %7.0f GB/s (%.2f of 8 TB/s)" % (2 * n * 8 / 2**30, 2 * n * 8 / t / 1e9, 2 * n * 8 / t / 8e12))
t = timed(lambda: x.sum())
print("read   %6.2f GiB read    : %7.0f GB/s (%.2f of 8 TB/s)" % (n * 8 / 2**30, n * 8 / t / 1e9, n * 8 / t / 8e12))
