#!/usr/bin/env python3
"""Follow-up to ab_placement.py (C3, 16 384 instances): are P and G faster to write when they lie far
apart in device memory?  Sets whose P and G are allocated with `spacer` GB of other allocations between them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import engine, problems  # noqa: E402

B = 16384
api = problems.load_api("mpc_interface")
get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
form = problems.lipm3d(api, N=32)
taus = np.random.default_rng(1).uniform(0.08, 0.12, B)
A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
asm = engine.Assembler(form, batch=B, lti=["LIP"])
asm.bind_lti("LIP", A, Bm)
given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [B, form.given_len]), device="cuda")
no, nc = asm.no, asm.nc
f = dict(dtype=torch.float64, device="cuda")


def timed(out, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    asm.assemble(given, out=out)
    e0.record()
    for _ in range(reps):
        asm.assemble(given, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


keep = []
for spacer_gb in (0, 0, 0, 0, 8, 8, 8, 8, 32, 32, 32, 32, 64, 64):
    P = torch.empty((B, no, no), **f)
    q = torch.empty((B, no), **f)
    try:
        sp = torch.empty(spacer_gb << 27, **f) if spacer_gb else None     # (2^27 doubles = 1 GB)
    except RuntimeError:                                                  # (the device is full)
        break
    G = torch.empty((B, nc, no), **f)
    h = torch.empty((B, nc), **f)
    keep.append((P, q, G, h, sp))
    t = min(timed((P, q, G, h)) for _ in range(3))
    print("spacer %3d GB: P at %x, G at %x (%.1f GB below)  %.3f ms" % (
        spacer_gb, P.data_ptr(), G.data_ptr(), (P.data_ptr() - G.data_ptr()) / 2**30, t))
