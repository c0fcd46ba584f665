#!/usr/bin/env python3
"""K1 (mpcasm_fill_su) throughput against the HBM roofline for the BASELINE shapes.

Algorithmic bytes per system: 8 (N n^2 + m N^2 n) written +
8 (n^2 + n m) read (LTI) or 8 N (n^2 + n m) (LTV)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import engine  # noqa: E402

CASES = [  # name, n, m, N, ltv, batch
    ("C2 biped LIPM", 3, 1, 16, False, 4096),
    ("C2 biped LIPM", 3, 1, 16, False, 8192),
    ("C2 biped LIPM", 3, 1, 16, False, 65536),
    ("C2 biped LIPM", 3, 1, 16, False, 524288),
    ("C3 N=32", 3, 1, 32, False, 131072),
    ("N=100 LTI", 3, 1, 100, False, 16384),
    ("C4 nx=12 nu=6 N=64", 12, 6, 64, False, 1024),
    ("C4 nx=12 nu=6 N=64", 12, 6, 64, False, 4096),
    ("C5 LTV N=100", 3, 1, 100, True, 2048),
    ("C5 LTV N=100", 3, 1, 100, True, 16384),
]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # substring of the case names to run
    rng = np.random.default_rng(0)
    print("%-22s %8s %10s %10s %9s %7s" % ("case", "batch", "MB/launch", "us/launch", "GB/s", "frac"))
    for name, n, m, N, ltv, batch in CASES:
        if only not in name:
            continue
        shapeA = (batch, N, n, n) if ltv else (batch, n, n)
        shapeB = (batch, N, n, m) if ltv else (batch, n, m)
        A = torch.as_tensor(rng.standard_normal(shapeA) / np.sqrt(n) * 0.9, device="cuda")
        B = torch.as_tensor(rng.standard_normal(shapeB), device="cuda")
        S = torch.empty((batch, N, n, n), dtype=torch.float64, device="cuda")
        U = torch.empty((batch, m, N, N, n), dtype=torch.float64, device="cuda")
        us = bench._event_ms(torch, lambda: engine.fill_su(A, B, N, ltv=ltv, out=(S, U)), 20) * 1e3
        nbytes = 8 * (N * n * n + m * N * N * n) + 8 * (n * n + n * m) * (N if ltv else 1)
        gbs = nbytes * batch / (us * 1e-6) / 1e9
        print("%-22s %8d %10.1f %10.1f %9.0f %7.3f" % (name, batch, nbytes * batch / 1e6, us, gbs,
                                                      gbs / 8000.0))
        del A, B, S, U
        torch.cuda.empty_cache()
    if not only:
        cpu_lines(rng)


def cpu_lines(rng):
    """The same fill on one host core -- the CPU baselines beside K1.
    The oracle is test infrastructure: its timing lives under tests/ and runs as a child."""
    import subprocess

    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cpu_fill_baseline.py")], check=True)


if __name__ == "__main__":
    main()
