#!/usr/bin/env python3
"""C4 (random LTI nx=12 nu=6, N=64) on the tiled kernel with horizon tables generated from
per-instance (A, B): the program the profiling scripts wrap.
   python tools/run_tiled_only.py [batch] [reps] [lti: 1 | 0] [what: all | cost | constraints] [MPCASM_OPT_PATH]
(path 0: the scan form where the plan has one -- with lti=0 the shared-model form --, 4: the Toeplitz form on the
matrix core, 3: the general form)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import capi, engine, problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lti = int(sys.argv[3]) if len(sys.argv) > 3 else 1
what = sys.argv[4] if len(sys.argv) > 4 else "all"
path = int(sys.argv[5]) if len(sys.argv) > 5 else 0
api = problems.load_api("mpc_interface")
rng = np.random.default_rng(20262)
form = problems.random_lti(api, rng, nx=12, nu=6, N=64)
asm = engine.Assembler(form, batch=batch, lti=["plant"] if lti else ())
if lti:
    As, Bs = zip(*(problems.random_lti_matrices(rng, 12, 6) for _ in range(batch)))
    asm.bind_lti("plant", torch.as_tensor(np.stack(As), device="cuda"),
                 torch.as_tensor(np.stack(Bs), device="cuda"))
asm.set_option(capi.OPT_PATH, path)
w = rng.uniform(0.1, 1.0, [batch, 1, 1])
asm.set_param("cost", "track s0", "weight", w)
given = torch.as_tensor(rng.normal(0, 0.3, [batch, form.given_len]), device="cuda")
kw = dict(want_cost=what != "constraints", want_constraints=what != "cost")
asm.assemble(given, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    asm.assemble(given, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
no, nc = asm.no, asm.nc
out_bytes = 8 * (no * no + no + nc * no + nc)
print("C4 N=64 tiled (lti=%d, %s, path %d: %s): %d instances in %.3f ms  %.3e assemblies/s  %.0f GB/s of output (%.3f of 8 TB/s)"
      % (lti, what, path, asm.last_kernel().split(" ")[0], batch, ms, batch / ms * 1e3, out_bytes * batch / ms / 1e6,
         out_bytes * batch / ms / 1e6 / 8000))
