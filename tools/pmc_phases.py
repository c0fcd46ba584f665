#!/usr/bin/env python3
"""Instruction counts of the persistent assembly kernel PHASE BY PHASE: the C2 assembly launched
with phases of the kernel switched off (MPCASM_OPT_PHASE_MASK; results are wrong then, only the
counters matter), two launches per mask, in the order of MASKS.  Run under
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS -- python3 tools/pmc_phases.py
and read the counter rows of resident_assemble_kernel in dispatch order (tools/pmc_phases.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import capi  # noqa: E402

MASKS = [("all", 0x3F), ("none", 0), ("staging", 0x10), ("compose", 1), ("hessian", 2),
         ("constraints", 8), ("Pq-store", 0x20)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    work = bench.build_workload(B, 1)
    asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
    asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
    given = torch.as_tensor(work["given"], device="cuda")
    lib = capi.load()
    for name, mask in MASKS:
        lib.mpcasm_set_option(capi.OPT_PHASE_MASK, mask | 0x80)
        for _ in range(2):
            asm.assemble(given)
        torch.cuda.synchronize()
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT)


if __name__ == "__main__":
    main()
