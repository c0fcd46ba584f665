#!/usr/bin/env python3
"""C4 (random LTI nx=12 nu=6, N=64: no=384, nc=1536) at its per-GPU batch of 8192 instances,
as `chunks` launches of `batch` instances on the staged pipeline -- the program
tools/profile_c4.sh wraps in rocprofv3.   python tools/run_c4_only.py [batch] [chunks] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import engine, problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
api = problems.load_api("mpc_interface")
form = problems.random_lti(api, np.random.default_rng(20262), nx=12, nu=6, N=64)
asm = engine.Assembler(form, batch=batch)
rng = np.random.default_rng(0)
given = [torch.as_tensor(rng.normal(0, 0.3, [batch, form.given_len]), device="cuda") for _ in range(chunks)]
asm.assemble(given[0])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    for g in given:
        asm.assemble(g)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
no, nc, ng = asm.no, asm.nc, asm.ng
R = sum(t[2] for t in asm.plan.itab[asm.plan.itab[22]:].reshape(-1)[:0]) if False else None
out_bytes = 8 * (no * no + no + nc * no + nc)
n = batch * chunks
print("C4 N=64: %d instances as %d x %d: %.2f ms  %.3e assemblies/s  %.0f GB/s of output (%.3f of 8 TB/s)"
      % (n, chunks, batch, ms, n / ms * 1e3, out_bytes * n / ms / 1e6, out_bytes * n / ms / 1e6 / 8000))
