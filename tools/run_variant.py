#!/usr/bin/env python3
"""One of bench.py's C2 variants by itself (for rocprofv3): run_variant.py 34|36 [reduced] [launches]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    wide = sys.argv[1] if len(sys.argv) > 1 else "34"
    reduced = "reduced" in sys.argv[2:]
    launches = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 2 else 2000
    dev = torch.device("cuda", 0)
    if os.environ.get("MPCASM_JIT"):
        from mpcasm import capi
        capi.load().mpcasm_set_option(capi.OPT_JIT, int(os.environ["MPCASM_JIT"]))
    B = 4096
    work = bench.build_workload(B, 20260, step_times=(7, 15) if wide == "34" else (6, 14), reduced=reduced)
    engine, form = work["engine"], work["form"]
    asm = engine.Assembler(form, batch=B, device=dev, lti=["LIP"])
    asm.bind_lti("LIP", torch.as_tensor(work["A"], device=dev), torch.as_tensor(work["B"], device=dev))
    asm.set_param("cost", "track vel_x", "aim", work["aims"])
    given = torch.as_tensor(work["given"], device=dev)
    f = dict(dtype=torch.float64, device=dev)
    sets = [(torch.empty((B, asm.no, asm.no), **f), torch.empty((B, asm.no), **f),
             torch.empty((B, asm.nc, asm.no), **f), torch.empty((B, asm.nc), **f)) for _ in range(4)]
    k = [0]

    def step():
        asm.assemble(given, out=sets[k[0] % 4])
        k[0] += 1

    if os.environ.get("MPCASM_ABLATE"):            # timing-only: phases of the kernel switched off
        from mpcasm import capi
        lib = capi.load()
        for name, off in (("all on", 0), ("no G", 8), ("no trips / P", 2), ("no q / zero blocks", 32),
                          ("no compose", 1), ("no G, P, q", 8 | 2 | 32), ("G only", 1 | 2 | 32),
                          ("G only, not stored", 1 | 2 | 32 | (512 << 16)), ("G not stored", 512 << 16),
                          ("runs of 2 instances", (1 << 10) << 16), ("runs of 4", (2 << 10) << 16),
                          ("runs of 8", (3 << 10) << 16), ("XCD-wise", 8192 << 16),
                          ("XCD-wise, runs of 8", (8192 | (3 << 10)) << 16), ("all on again", 0)):
            lib.mpcasm_set_option(capi.OPT_PHASE_MASK, (capi.PHASE_DEFAULT & ~(off & 0xFFFF)) | (off >> 16))
            print("  no=%d %-20s %.2f us" % (asm.no, name, bench._event_ms(torch, step, launches) * 1e3))
        lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT)
        return
    ms = bench._event_ms(torch, step, launches)
    print("no=%d nc=%d %s: %.2f us per launch (hipEvents, %d launches), kernel %s"
          % (asm.no, asm.nc, "reduced" if reduced else "full", ms * 1e3, launches, asm.last_kernel()))


if __name__ == "__main__":
    main()
