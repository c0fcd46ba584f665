#!/usr/bin/env python3
"""How P leaves the persistent kernel, A/B on one box: MPCASM_OPT_P_DIRECT = 1 (blocks straight to
HBM) against 2 (collected in LDS) on the biped at a given horizon and batch, device warm."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import capi, engine, problems  # noqa: E402


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 12          # step_samples: N = 2 x this
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    api = problems.load_api("mpc_interface")
    form = problems.biped(api, problems.BipedConfig(step_samples=samples))
    form.update(step_times=np.array([samples - 2, 2 * samples - 2]), step_count=0)
    lib = capi.load()
    given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [batch, form.given_len]), device="cuda")
    for mode in (0, 1, 2):
        lib.mpcasm_set_option(capi.OPT_P_DIRECT, mode)
        asm = engine.Assembler(form, batch=batch)
        lib.mpcasm_set_option(capi.OPT_P_DIRECT, 0)
        ms = bench._event_ms(torch, lambda: asm.assemble(given), 30)
        out = 8 * (asm.no * asm.no + asm.no + asm.nc * asm.no + asm.nc)
        print("N=%d B=%d P_DIRECT=%d: %.3f ms  %.3e/s  %.3f of 8 TB/s" % (2 * samples, batch, mode, ms, batch / ms * 1e3,
                                                                     out * batch / ms / 1e6 / 8000))


if __name__ == "__main__":
    main()
