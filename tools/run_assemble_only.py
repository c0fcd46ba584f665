#!/usr/bin/env python3
"""A few launches of the C2 assembly alone (for rocprofv3 --pmc runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
work = bench.build_workload(B, 1)
lti = os.environ.get("MPCASM_LTI") == "1"      # horizon matrices generated on chip
asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"] if lti else ())
if lti:
    asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
given = torch.as_tensor(work["given"], device="cuda")
for _ in range(6):
    asm.assemble(given)
torch.cuda.synchronize()
