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


def box_store_rate():
    """TB/s of a plain 512 MB fill on this box: boxes differ by several per cent, so times from
    different calls are compared relative to this."""
    x = torch.empty(1 << 26, dtype=torch.float64, device="cuda")
    for _ in range(3):
        x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        x.fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return x.numel() * 8 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12


def main():
    lib = capi.load()
    print("box: 512 MB fill at %.2f TB/s" % box_store_rate())
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
        # outputs rotate over three sets, as in bench.py (a set is not rewritten while its
        # lines still sit in the Infinity Cache)
        sets = [tuple(torch.empty_like(t) for t in asm.assemble(given)) for _ in range(3)]
        turn = [0]

        def step():
            turn[0] = (turn[0] + 1) % 3
            asm.assemble(given, out=sets[turn[0]])

        if "MPCASM_PHASES" in os.environ:       # e.g. 0x1BF: instances one by one round the workgroups
            lib.mpcasm_set_option(capi.OPT_PHASE_MASK, int(os.environ["MPCASM_PHASES"], 0))
        res = {2: [], 1: []}
        for rnd in range(5):
            for mode in (2, 1):
                lib.mpcasm_set_option(capi.OPT_JIT, mode)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    step()
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
        print("B=%6d  ahead-of-time %8.1f us (min %8.1f)   per-plan %8.1f us (min %8.1f)"
              % (B, np.median(res[2]), min(res[2]), np.median(res[1]), min(res[1])))


if __name__ == "__main__":
    main()
