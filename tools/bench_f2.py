#!/usr/bin/env python3
"""f2 alone (rows of every definition straight from the sources, mpcasm_preview_direct) at several batch
sizes, hipEvents around back-to-back calls: the kernel's rate once the host is not the limit.
bench_f2.py [batch ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import engine  # noqa: E402

for B in [int(x) for x in sys.argv[1:]] or [4096, 65536]:
    work = bench.build_workload(B, 1)
    form = work["form"]
    given = torch.as_tensor(work["given"], device="cuda")
    for label, kw in (("shared S, U", {}), ("tables from per-instance (A, B)", dict(lti=["LIP"]))):
        asm = engine.Assembler(form, batch=B, **kw)
        if kw:
            asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
        optim = torch.as_tensor(np.random.default_rng(1).normal(0, 0.5, [B, asm.no]), device="cuda")
        rows = asm.preview_rows(given, optim)
        ms = bench._event_ms(torch, lambda: asm.preview_rows(given, optim, out=rows), 200)
        algo = 8 * (asm.ng + asm.no + asm.plan.pmrows) + (8 * 12 if kw else 0)
        print("f2 rows, %-32s B=%6d x %d rows  %8.4f ms  %.3e instances/s  %6.0f GB/s = %.3f of 8 TB/s"
              % (label + ":", B, asm.plan.pmrows, ms, B / ms * 1e3, algo * B / ms / 1e6, algo * B / ms / 1e6 / 8000))
        del asm
