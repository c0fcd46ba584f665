#!/usr/bin/env python3
"""C3 (3-D LIPM N=32: no=96, nc=196) at B=16384 -- for rocprofv3.  python tools/run_c3_only.py [batch] [reps] [lti]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mpcasm import engine, problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lti = len(sys.argv) > 3 and sys.argv[3] == "lti"
from mpcasm import capi  # noqa: E402

if os.environ.get("MPCASM_PER_CU"):                      # workgroups per CU of the persistent kernel
    capi.load().mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, int(os.environ["MPCASM_PER_CU"]))
if os.environ.get("MPCASM_PHASES"):                      # timing-only ablation (tools/run_variant.py)
    capi.load().mpcasm_set_option(capi.OPT_PHASE_MASK, int(os.environ["MPCASM_PHASES"], 0))
api = problems.load_api("mpc_interface")
form = problems.lipm3d(api, N=32)
asm = engine.Assembler(form, batch=batch, lti=["LIP"] if lti else ())
given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [batch, form.given_len]), device="cuda")
asm.assemble(given)
torch.cuda.synchronize()
import time  # noqa: E402

t0 = time.perf_counter()                      # the clocks settle (tools/launch_series.py)
while (time.perf_counter() - t0) * 1e3 < 40.0:
    for _ in range(4):
        asm.assemble(given)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    asm.assemble(given)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
no, nc = asm.no, asm.nc
out_bytes = 8 * (no * no + no + nc * no + nc)
print("C3 N=32 B=%d%s: %.3f ms  %.3e assemblies/s  %.0f GB/s of output (%.3f of 8 TB/s)"
      % (batch, " (K1 fused)" if lti else "", ms, batch / ms * 1e3, out_bytes * batch / ms / 1e6,
         out_bytes * batch / ms / 1e6 / 8000))
