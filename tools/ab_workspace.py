#!/usr/bin/env python3
"""Same box, one process, interleaved rounds: C3 (3-D LIPM N=32, horizon matrices on chip) and the biped
at N=24 with the persistent kernel's workspace compact or dense (MPCASM_NO_COMPACT) and P leaving
the kernel by the shipped rule, directly or through LDS (MPCASM_OPT_P_DIRECT).  ab_workspace.py [batch] [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import capi, engine, problems  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
api = problems.load_api("mpc_interface")


def timed(asm, given, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        asm.assemble(given)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(name, form, lti_ab):
    given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [B, form.given_len]), device="cuda")
    variants = {}
    for label, env in (("as shipped", {}), ("compact, P direct", {"P_DIRECT": "1", "WORKSPACE": "compact"}),
                       ("compact, P via LDS", {"P_DIRECT": "2", "WORKSPACE": "compact"}),
                       ("dense, P direct", {"P_DIRECT": "1", "WORKSPACE": "dense"}),
                       ("dense, P via LDS", {"P_DIRECT": "2", "WORKSPACE": "dense"})):
        os.environ.pop("MPCASM_NO_COMPACT", None)
        env = dict(env)
        capi.load().mpcasm_set_option(capi.OPT_P_DIRECT, int(env.pop("P_DIRECT", "0")))
        os.environ.update(env)
        asm = engine.Assembler(form, batch=B, lti=["LIP"] if lti_ab else [],
                               workspace=env.pop("WORKSPACE", "auto"))
        if lti_ab:
            asm.bind_lti("LIP", *lti_ab)
        asm.assemble(given)
        variants[label] = asm
    os.environ.pop("MPCASM_NO_COMPACT", None)
    capi.load().mpcasm_set_option(capi.OPT_P_DIRECT, 0)
    torch.cuda.synchronize()
    times = {k: [] for k in variants}
    for r in range(rounds + 1):
        for k, asm in variants.items():
            ms = timed(asm, given, 40)
            if r:                                   # (round 0: the clocks settle)
                times[k].append(ms)
    asm = next(iter(variants.values()))
    out = 8 * (asm.no ** 2 + asm.no + asm.nc * asm.no + asm.nc)
    for k, v in times.items():
        med = float(np.median(v))
        print("%-16s %-28s B=%d  median %.3f ms (min %.3f)  %.3e asm/s  %.3f of 8 TB/s"
              % (name, k, B, med, min(v), B / med * 1e3, out * B / med / 1e9 / 8))


get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
taus = np.random.default_rng(1).uniform(0.08, 0.12, B)
A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
case("C3 lipm3d N=32", problems.lipm3d(api, N=32), (A, Bm))
biped = problems.biped(api, problems.BipedConfig(step_samples=12))
biped.update(step_times=np.array([10, 22]), step_count=0)
case("biped N=24", biped, (A, Bm))
case("biped N=24, S U read", biped, None)
