#!/usr/bin/env python3
"""For rocprofv3 --pmc (tools/pmc_placement.sh): C3 at 16 384 instances into several separate allocations
of its result arrays; the LAST six launches of the process go three times into the fastest set and three
times into the slowest (tools/ab_placement.py: where the results lie decides a third of the time)."""
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
sets = [(torch.empty((B, no, no), **f), torch.empty((B, no), **f), torch.empty((B, nc, no), **f),
         torch.empty((B, nc), **f)) for _ in range(int(os.environ.get("MPCASM_NSETS", "10")))]


def timed(out, reps=4):
    asm.assemble(given, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        asm.assemble(given, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = [min(timed(s) for _ in range(2)) for s in sets]
fast, slow = int(np.argmin(ms)), int(np.argmax(ms))
print("sets: " + " ".join("%.3f" % t for t in ms))
print("fastest %d (%.3f ms), slowest %d (%.3f ms)" % (fast, ms[fast], slow, ms[slow]))
torch.cuda.synchronize()
for k in (fast, fast, fast, slow, slow, slow):
    asm.assemble(given, out=sets[k])
torch.cuda.synchronize()
