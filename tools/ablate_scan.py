#!/usr/bin/env python3
"""Timing-only ablation of the tiled kernel's scan form on C4 (MPCASM_OPT_PHASE_MASK: results are WRONG
with a phase off): what the row blocks of P, the gradient, the rows of G and the stores of P account for.
   python tools/ablate_scan.py [batch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import capi, engine, problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
api = problems.load_api("mpc_interface")
rng = np.random.default_rng(20262)
form = problems.random_lti(api, rng, nx=12, nu=6, N=64)
asm = engine.Assembler(form, batch=batch, lti=["plant"])
As, Bs = zip(*(problems.random_lti_matrices(rng, 12, 6) for _ in range(batch)))
asm.bind_lti("plant", torch.as_tensor(np.stack(As), device="cuda"), torch.as_tensor(np.stack(Bs), device="cuda"))
given = torch.as_tensor(rng.normal(0, 0.3, [batch, form.given_len]), device="cuda")
lib = capi.load()


def timed(mask, reps=5):
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, mask)
    asm.assemble(given)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            asm.assemble(given)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


full = 0xBF
rows = [("everything", full), ("no row blocks of P", full & ~2), ("no gradient", full & ~4),
        ("no rows of G, no h", full & ~8), ("P computed, not stored", full & ~32),
        ("only rows of G + h", full & ~(2 | 4)), ("only P", full & ~(4 | 8)), ("only the gradient", full & ~(2 | 8)),
        ("only set-up", full & ~(2 | 4 | 8))]
try:
    for name, mask in rows:
        ms = timed(mask)
        print("%-26s %8.3f ms   (%s)" % (name, ms, asm.last_kernel().split(" ")[0]))
finally:
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, full)
