#!/usr/bin/env python3
"""Persistent kernel: time per launch against batch size and workgroups per CU
(MPCASM_OPT_RESIDENT_PER_CU), C2 with the horizon matrices built on chip."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import capi  # noqa: E402


def main():
    lib = capi.load()
    for B in (4096, 16384):
        work = bench.build_workload(B, 1)
        asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
        asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
        given = torch.as_tensor(work["given"], device="cuda")
        line = "B=%6d " % B
        for k in (1, 2):
            lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, k)
            ts = [bench._event_ms(torch, lambda: asm.assemble(given), 60) * 1e3]   # (device warm)
            line += "  k=%d %8.1f us" % (k, np.median(ts))
        lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
