#!/usr/bin/env python3
"""biped N=24 (the example as shipped) with its horizon matrices built on chip, B=16384: how P leaves
(MPCASM_OPT_P_DIRECT) decides whether one or two workgroups share a CU (85 against 64 KB of LDS)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import capi, engine, problems  # noqa: E402

api = problems.load_api("mpc_interface")
lib = capi.load()
batches = [int(x) for x in sys.argv[1:]] or [16384]
for B, (N2, times) in [(b, nt) for nt in ((12, (10, 22)), (8, (6, 14))) for b in batches]:
    form = problems.biped(api, problems.BipedConfig(step_samples=N2))
    form.update(step_times=np.array(times), step_count=0)
    get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
    taus = np.random.default_rng(1).uniform(0.08, 0.12, B)
    A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
    Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
    given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [B, form.given_len]), device="cuda")
    for opt in (1, 2):
        lib.mpcasm_set_option(capi.OPT_P_DIRECT, opt)
        asm = engine.Assembler(form, batch=B, lti=["LIP"])
        asm.bind_lti("LIP", A, Bm)
        ms = bench._event_ms(torch, lambda: asm.assemble(given), 20)
        out = 8 * (asm.no ** 2 + asm.no + asm.nc * asm.no + asm.nc)
        print("N=%d B=%d P_DIRECT=%d  %.3f ms  %.3e asm/s  %.2f TB/s (%.3f)" % (
            2 * N2, B, opt, ms, B / ms * 1e3, out * B / ms / 1e9, out * B / ms / 1e9 / 8))
        del asm
lib.mpcasm_set_option(capi.OPT_P_DIRECT, 0)
