#!/usr/bin/env python3
"""Stress the persistent kernel's instance loop: random formulation structures, batches of
several instances per workgroup (double-buffered images, shared-element clearing, P / q
reuse across instances), every instance compared with the staged pipeline's result.

    python tools/stress_resident.py [first_seed] [last_seed] [batch]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpc-interface_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from mpcasm import capi, problems  # noqa: E402
from mpcasm.engine import Assembler  # noqa: E402
from test_random_formulations import random_formulation  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1800
    api = problems.load_api("mpc_interface")
    lib = capi.load()
    worst, resident = 0.0, 0
    for seed in range(first, last):
        form, rng = random_formulation(api, seed)
        given = rng.standard_normal([batch, form.given_len])
        asm = Assembler(form, batch=batch)
        # per-instance parameters too: weights and extremes differ between instances
        asm.params[:] = asm.params * torch.as_tensor(
            rng.uniform(0.5, 1.5, tuple(asm.params.shape)), device=asm.params.device)
        out = {}
        for path in (2, 0):
            lib.mpcasm_set_option(capi.OPT_PATH, path)
            out[path] = [t.cpu().numpy().copy() for t in asm.assemble(given)]
        lib.mpcasm_set_option(capi.OPT_PATH, 0)
        resident += int(asm.plan.resident["ok"])
        for a, b, name in zip(out[0], out[2], "PqGh"):
            scale = max(1.0, float(np.abs(b).max()))
            err = float(np.abs(a - b).max()) / scale
            worst = max(worst, err)
            if not err < 1e-12:
                bad = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
                print("seed %d %s: relative error %.3e at %s (instance %d of %d)"
                      % (seed, name, err, bad, bad[0], batch))
                sys.exit(1)
    print("seeds %d..%d, batch %d: %d on the persistent kernel, worst relative difference %.2e"
          % (first, last - 1, batch, resident, worst))


if __name__ == "__main__":
    main()
