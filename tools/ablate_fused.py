#!/usr/bin/env python3
"""Timing-only ablation of the fused assembly kernel (profiling aid).

Times mpcasm_assemble on the C2 biped workload with phases of the fused kernel
switched off through MPCASM_OPT_PHASE_MASK (results are wrong then; only the
time matters), interleaved rounds in one process (cdna guide section 5.4 rule 24).
"""
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    work = bench.build_workload(B, 1)
    engine, form = work["engine"], work["form"]
    lti = os.environ.get("MPCASM_LTI") == "1"   # horizon matrices generated on chip
    asm = engine.Assembler(form, batch=B, lti=["LIP"] if lti else ())
    if lti:
        asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
    given = torch.as_tensor(work["given"], device="cuda")
    lib = capi.load()
    if os.environ.get("MPCASM_JIT"):
        lib.mpcasm_set_option(capi.OPT_JIT, int(os.environ["MPCASM_JIT"]))
    lib.mpcasm_set_option(capi.OPT_PATH, int(os.environ.get("MPCASM_PATH", "0")))
    pf = 0x80 if os.environ.get("MPCASM_PREFETCH", "1") == "1" else 0   # register prefetch
    masks = [("all", 0x3F), ("none", 0), ("staging", 0x10), ("compose", 1), ("hessian", 2),
             ("gradient", 4), ("constraints", 8), ("Pq-store", 0x20), ("no-staging", 0x2F),
             ("no-compose", 0x3E), ("no-hessian", 0x3D), ("no-grad", 0x3B), ("no-constr", 0x37),
             ("no-Pstore", 0x1F), ("no-grad-constr", 0x33), ("no-hess-grad", 0x39)]
    masks = [(name, mask | pf) for name, mask in masks]
    times = {name: [] for name, _ in masks}
    for rnd in range(6):
        for name, mask in masks:
            lib.mpcasm_set_option(capi.OPT_PHASE_MASK, mask)
            for _ in range(3):
                asm.assemble(given)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                asm.assemble(given)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT)
    print("B=%d  no=%d nc=%d  (us per launch: median / min)" % (B, asm.no, asm.nc))
    for name, _ in masks:
        t = np.array(times[name])
        print("%-14s %8.1f %8.1f" % (name, np.median(t), t.min()))


if __name__ == "__main__":
    main()
