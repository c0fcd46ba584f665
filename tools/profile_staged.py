#!/usr/bin/env python3
"""Run the staged pipeline on C3 / C4 (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import engine, problems  # noqa: E402

api = problems.load_api("mpc_interface")
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    form, batch = problems.lipm3d(api, N=32), 16384
else:
    form, batch = problems.random_lti(api, np.random.default_rng(20262), N=64), 256
asm = engine.Assembler(form, batch=batch)
given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [batch, form.given_len]), device="cuda")
for _ in range(5):
    asm.assemble(given)
torch.cuda.synchronize()
