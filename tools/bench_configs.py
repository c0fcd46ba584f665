#!/usr/bin/env python3
"""Assembly throughput of the BASELINE configurations C2, C3, C4 on
whatever path the library selects for them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import engine, problems  # noqa: E402


def run(name, form, batch, reps=10, lti=None):
    rng = np.random.default_rng(0)
    asm = engine.Assembler(form, batch=batch, lti=[lti[0]] if lti else ())
    if lti:     # horizon matrices built on chip from per-instance (A, B)
        asm.bind_lti(lti[0], torch.as_tensor(lti[1], device="cuda"), torch.as_tensor(lti[2], device="cuda"))
    given = torch.as_tensor(rng.normal(0, 0.1, [batch, form.given_len]), device="cuda")
    ms = bench._event_ms(torch, lambda: asm.assemble(given), reps)   # (after the clocks have settled)
    no, nc, ng = asm.no, asm.nc, asm.ng
    out_bytes = 8 * (no * no + no + nc * no + nc)
    print("%-28s B=%6d no=%4d nc=%5d rtot=%5d  %9.3f ms  %10.0f asm/s  %7.1f GB/s out"
          % (name, batch, no, nc, asm.plan.rtot, ms, batch / ms * 1e3, out_bytes * batch / ms / 1e6))


def main():
    api = problems.load_api("mpc_interface")
    form = problems.biped(api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    run("C2 biped N=16", form, 4096)
    run("C2 biped N=16", form, 65536)
    form = problems.biped(api, problems.BipedConfig(step_samples=12))
    form.update(step_times=np.array([10, 22]), step_count=0)
    run("biped N=24 (as shipped)", form, 16384)
    get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
    taus = np.random.default_rng(1).uniform(0.08, 0.12, 16384)
    A = np.stack([get_A(tau=t) for t in taus])
    B = np.stack([get_B(tau=t) for t in taus])
    run("biped N=24, K1 fused", form, 16384, lti=("LIP", A, B))
    form3 = problems.lipm3d(api, N=32)
    run("C3 lipm3d N=32 (staged)", form3, 16384)
    lip3 = form3.dynamics["LIP"]
    # S[0][j][i] = A[i][j], U_0[0][0][i] = B[i][0]  (tools.py:14-33): the system's own (A, B)
    run("C3 lipm3d N=32, K1 fused", form3, 16384,
        lti=("LIP", np.broadcast_to(lip3.matrices[1][0].T, (16384, 3, 3)).copy(),
             np.broadcast_to(lip3.matrices[0][0, 0, :, None], (16384, 3, 1)).copy()))
    run("C4 random LTI N=64", problems.random_lti(api, np.random.default_rng(20262), N=64), 256, 3)


if __name__ == "__main__":
    main()
