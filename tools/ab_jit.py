#!/usr/bin/env python3
"""A/B on one box: the C2 assembly on the ahead-of-time persistent kernel and on the kernel
compiled for the plan (MPCASM_OPT_JIT), interleaved rounds in one process."""
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
    for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
        work = bench.build_workload(min(B, 4096), 1)
        times = (B + 4095) // 4096
        tile = lambda x: np.concatenate([x] * times)[:B]
        lib.mpcasm_set_option(capi.OPT_P_DIRECT, int(os.environ.get("MPCASM_P_DIRECT", "0")))
        asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
        lib.mpcasm_set_option(capi.OPT_P_DIRECT, 0)
        asm.bind_lti("LIP", torch.as_tensor(tile(work["A"]), device="cuda"),
                     torch.as_tensor(tile(work["B"]), device="cuda"))
        given = torch.as_tensor(tile(work["given"]), device="cuda")
        res = {2: [], 1: []}
        for rnd in range(5):
            for mode in (2, 1):
                lib.mpcasm_set_option(capi.OPT_JIT, mode)
                for _ in range(3):
                    asm.assemble(given)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    asm.assemble(given)
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
        print("B=%6d  ahead-of-time %8.1f us (min %8.1f)   per-plan %8.1f us (min %8.1f)"
              % (B, np.median(res[2]), min(res[2]), np.median(res[1]), min(res[1])))


if __name__ == "__main__":
    main()
