#!/usr/bin/env python3
"""One K1 case, launched `reps` times -- the program rocprofv3 wraps in tools/profile_fill.sh.

    python tools/run_fill_only.py n m N ltv batch [reps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))

import torch  # noqa: E402

from mpcasm import engine  # noqa: E402

n, m, N, ltv, batch = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
rng = np.random.default_rng(0)
shapeA = (batch, N, n, n) if ltv else (batch, n, n)
shapeB = (batch, N, n, m) if ltv else (batch, n, m)
A = torch.as_tensor(rng.standard_normal(shapeA) / np.sqrt(n) * 0.9, device="cuda")
B = torch.as_tensor(rng.standard_normal(shapeB), device="cuda")
S = torch.empty((batch, N, n, n), dtype=torch.float64, device="cuda")
U = torch.empty((batch, m, N, N, n), dtype=torch.float64, device="cuda")
import time  # noqa: E402

t0 = time.perf_counter()                      # the clocks settle (tools/launch_series.py) ...
while (time.perf_counter() - t0) * 1e3 < 40.0:
    for _ in range(4):
        engine.fill_su(A, B, N, ltv=bool(ltv), out=(S, U))
    torch.cuda.synchronize()
for _ in range(reps):                         # ... these are the launches the summary reads
    engine.fill_su(A, B, N, ltv=bool(ltv), out=(S, U))
torch.cuda.synchronize()
