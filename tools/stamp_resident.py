#!/usr/bin/env python3
"""In-kernel cycle stamps of the persistent assembly kernel (diagnostic build path,
MPCASM_OPT_PHASE_MASK bit 6): where a workgroup's time per instance goes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import capi  # noqa: E402

NAMES = ["barA", "compose+B", "tiles", "G,h", "barC", "dma wait", "Pq out", "fetch"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    times = os.environ.get("MPCASM_STEP_TIMES")      # e.g. "7,15": the 34-wide bucket
    work = bench.build_workload(B, 1, step_times=[int(x) for x in times.split(",")] if times else (6, 14))
    engine, form = work["engine"], work["form"]
    lti = os.environ.get("MPCASM_LTI") == "1"   # horizon matrices generated on chip
    asm = engine.Assembler(form, batch=B, lti=["LIP"] if lti else ())
    if lti:
        asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
    given = torch.as_tensor(work["given"], device="cuda")
    lib = capi.load()
    if os.environ.get("MPCASM_JIT"):
        lib.mpcasm_set_option(capi.OPT_JIT, int(os.environ["MPCASM_JIT"]))
    if os.environ.get("MPCASM_PER_CU"):       # workgroups per CU (is the set-up bound by the CU's load path?)
        lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, int(os.environ["MPCASM_PER_CU"]))
    for _ in range(3):
        asm.assemble(given)
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT | capi.PHASE_STAMPS)
    asm._work.zero_()
    asm.assemble(given)
    torch.cuda.synchronize()
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT)
    raw = asm._work.view(torch.int64).cpu().numpy()
    n64 = (raw.size // 64) * 64
    # workgroups that stamped (phase sums, then as many rows of set-up stations)
    used = int((raw[:n64].reshape(-1, 64)[: 2 * 256 * 8].sum(axis=1) != 0).sum()) // 2
    grid = max(1, min(B, used))
    t = raw[:grid * 8 * 8].reshape(grid, 8, 8).astype(np.float64)
    per_wg = B / grid
    print("B=%d grid=%d  instances per workgroup %.2f; cycles per instance (100 MHz ticks x?)"
          % (B, grid, per_wg))
    setup_stamps(raw, grid)
    for w in range(8):
        row = t[:, w, :].mean(axis=0) / per_wg
        print("wave %d: " % w + "  ".join("%s %7.0f" % (n, v) for n, v in zip(NAMES, row))
              + "   total %8.0f" % row.sum())


def setup_stamps(raw, grid):
    """Cycles since kernel start at the stations of the once-per-workgroup set-up."""
    s = raw[grid * 64:2 * grid * 64].reshape(grid, 8, 8).astype(np.float64)
    names = ["stream tables", "barrier", "compose regs", "first tables", "tables copied", "barrier",
             "first image", "loop"]
    for w in (0, 1, 4, 7):
        print("set-up, wave %d: " % w + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, s[:, w, :].mean(axis=0))))


if __name__ == "__main__":
    main()
