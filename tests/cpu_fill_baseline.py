#!/usr/bin/env python3
"""K1 on one host core: the compiled C restatement (oracle/extend_matrices.c) and the numpy
oracle timed on the shapes of tools/bench_fill.py -- the CPU baselines beside the fill kernel
(SURVEY.md section 8d).  Not a test: run by tools/bench_fill.py, or by hand."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    from oracle import c_oracle, qp_oracle

    rng = np.random.default_rng(0)
    print("%-22s %8s %14s %14s" % ("CPU, one core", "systems", "C  systems/s", "numpy systems/s"))
    for name, n, m, N, count in (("C2 biped LIPM", 3, 1, 16, 20000), ("C4 nx=12 nu=6 N=64", 12, 6, 64, 60)):
        A = rng.standard_normal((count, n, n)) / np.sqrt(n) * 0.9
        B = rng.standard_normal((count, n, m))
        t0 = time.perf_counter()
        c_oracle.extend_matrices_batch(A, B, N)
        tc = time.perf_counter() - t0
        k = max(1, count // 20)
        t0 = time.perf_counter()
        for b in range(k):
            qp_oracle.extend_matrices(N, A[b], B[b])
        tn = (time.perf_counter() - t0) / k * count
        print("%-22s %8d %14.0f %14.0f" % (name, count, count / tc, count / tn))


if __name__ == "__main__":
    main()
