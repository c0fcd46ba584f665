#!/usr/bin/env python3
"""C5 as an assembly (problems.lipm_ltv: LTV LIPM, N = 100, two axes; 200 unknowns, 404 lines) on the sweep
kernel, per-step per-instance (A_k, B_k) -- beside the route it replaces: mpcasm_fill_su(ltv) writes S, U,
the staged / tiled assembly reads them back.   python tools/run_c5_only.py [batch] [reps] [route: sweep | fill]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import engine, problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
route = sys.argv[3] if len(sys.argv) > 3 else "sweep"
N = 100
api = problems.load_api("mpc_interface")
rng = np.random.default_rng(20263)
form = problems.lipm_ltv(api, N=N)
first = [problems.ltv_lipm_steps(api, N=N, theta=float(t)) for t in rng.uniform(0, 2 * np.pi, 8)]
A = torch.as_tensor(np.stack([first[i % 8][0] * (1.0 - 1e-3 * (i // 8) / max(batch // 8, 1)) for i in range(batch)]),
                    device="cuda")
Bm = torch.as_tensor(np.stack([first[i % 8][1] * (1.0 + 1e-3 * (i // 8) / max(batch // 8, 1)) for i in range(batch)]),
                     device="cuda")
given = torch.as_tensor(rng.normal(0, 0.05, [batch, form.given_len]), device="cuda")
if route == "sweep":
    asm = engine.Assembler(form, batch=batch, ltv=["LIP"])
    asm.bind_ltv("LIP", A, Bm)

    what = os.environ.get("MPCASM_C5_WHAT", "all")       # all | cost | constraints (timing-only ablation)
    kw = dict(want_cost=what != "constraints", want_constraints=what != "cost")

    def step():
        asm.assemble(given, **kw)
else:
    asm = engine.Assembler(form, batch=batch)
    S, U = engine.fill_su(A, Bm, N, ltv=True)

    def step():
        engine.fill_su(A, Bm, N, ltv=True, out=(S, U))
        asm.bind_source(("LIP", 0), U[:, 0], check=False)
        asm.bind_source(("LIP", 1), S)
        asm.assemble(given)
step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / reps)
no, nc = asm.no, asm.nc
out_bytes = 8 * (no * no + no + nc * no + nc)
in_bytes = 8 * (N * (9 + 3) + asm.ng + int(asm.params.shape[1]))
print("C5 assembly N=100 (%s: %s): %d instances in %.3f ms  %.3e assemblies/s  %.0f GB/s algorithmic (%.3f of 8 TB/s)"
      % (route, asm.last_kernel().split(" ")[0], batch, best, batch / best * 1e3,
         (out_bytes + in_bytes) * batch / best / 1e6, (out_bytes + in_bytes) * batch / best / 1e6 / 8000))
