"""GPU: compiled kernels -- ahead of the first launch (mpcasm_plan_prepare), kept on disk for the
next process, and once per plan structure over the per-part calls of the walking loop."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from mpcasm import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from mpcasm import capi, engine, problems
api = problems.load_api("mpc_interface")
form = problems.biped(api, problems.BipedConfig(step_samples=8))
form.update(step_times=np.array([6, 14]), step_count=0)
asm = engine.Assembler(form, batch=1024, lti=["LIP"])          # (prepare: compiles or loads here)
out = (ctypes.c_int64 * 3)()
capi.load().mpcasm_jit_stats(out)
before = list(out)
given = torch.zeros((1024, form.given_len), dtype=torch.float64, device="cuda")
asm.assemble(given)
asm.assemble(given, count=5)                                   # a small launch: the same kernel
small = asm.last_kernel()
torch.cuda.synchronize()
capi.load().mpcasm_jit_stats(out)
print("STATS", before[0], before[1], out[0], out[1], "hiprtc" in small)
"""


def _child(cache):
    env = dict(os.environ, MPCASM_CACHE_DIR=cache)
    proc = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "mpc-interface_amd"), ROOT)],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("STATS")][0].split()
    return [int(x) for x in line[1:5]] + [line[5] == "True"]


def test_a_second_process_compiles_nothing(gpu_api, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    cache = str(tmp_path / "co")
    first = _child(cache)
    if first[0] == 0 and first[1] == 0:
        pytest.skip("no libhiprtc.so on this box: the ahead-of-time kernel ran")
    # the Assembler compiled at creation (capacity 1024), the launches compiled nothing more,
    # and the 5-instance launch ran the compiled kernel as well
    assert first[0] >= 1 and first[2] == first[0] and first[4]
    second = _child(cache)
    assert second[0] == 0 and second[2] == 0 and second[1] >= 1 and second[4]


def test_a_per_part_sweep_compiles_each_plan_once(gpu_api):
    """generate_qp_cost / generate_qp_constraint for every part of the biped on every tick of
    two walking cycles: the plans of the three structures stay in the (least recently used)
    cache, a phase that comes back compiles nothing."""
    from mpcasm import plan as planmod

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    clock = problems.StepClock(8, form.domain["Ds_x"])
    rng = np.random.default_rng(0)
    calls = []
    real = planmod.compile_plan

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)

    import mpcasm.engine as engine

    engine_compile = engine.compile_plan
    engine.compile_plan = counting
    try:
        per_tick = []
        for tick in range(26):
            form.update(step_times=clock.step_times, step_count=clock.step_count)
            given = form.arrange_given(problems.biped_given_collector(form, rng))
            n0 = len(calls)
            for cost in form.goals.values():
                form.generate_qp_cost(cost, given)
            for limit in form._all_limits():
                form.generate_qp_constraint(limit, given)
            form.generate_all_qp_matrices(given)
            per_tick.append(len(calls) - n0)
            clock.tick()
    finally:
        engine.compile_plan = engine_compile
    parts = 1 + len(form.goals) + len(form._all_limits())
    assert per_tick[0] == parts                       # the first tick compiles every part once
    # later ticks compile only what a NEW structure needs; once the walking cycle has shown its
    # structures, nothing
    assert sum(per_tick) <= 3 * parts and sum(per_tick[17:]) == 0, per_tick
