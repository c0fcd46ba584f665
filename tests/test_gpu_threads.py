"""GPU: the launch path from several host threads ("safe to call from multiple Python
threads on different streams").  What used to be per-thread state of a per-function attribute -- the
dynamic-LDS limit of a kernel -- is raised once, process-wide (kernels.h allow_whole_lds): a thread that
launches a small-LDS plan of an instantiation cannot lower the limit under another thread's larger plan."""
import threading

import numpy as np
import pytest

from mpcasm import problems

pytestmark = pytest.mark.gpu


def test_two_host_threads_alternate_a_small_and_a_large_lds_plan(gpu_api):
    """Two threads, a stream each, 200 launches each of a 36-wide biped plan (35 KB of LDS, the kernel
    compiled for the plan) alternating with a C3 plan (3-D LIPM N=32: beyond 64 KB, the dynamic-LDS limit
    must have been raised): every result equal, bit for bit, to the single-threaded one."""
    import torch

    from mpcasm import engine

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    rng = np.random.default_rng(11)
    conf = problems.BipedConfig(step_samples=8)
    biped = problems.biped(gpu_api, conf)
    biped.update(step_times=np.array([6, 14]), step_count=0)
    lip3 = problems.lipm3d(gpu_api, N=32)
    cases = {"biped": (biped, 512), "c3": (lip3, 48)}
    given = {k: torch.as_tensor(rng.normal(0, 0.05, [b, f.given_len]), device="cuda") for k, (f, b) in cases.items()}

    def make():
        return {k: engine.Assembler(f, batch=b, lti=["LIP"]) for k, (f, b) in cases.items()}

    ref_asm = make()
    ref = {k: tuple(t.clone() for t in ref_asm[k].assemble(given[k])) for k in cases}
    assert "persistent" in ref_asm["c3"].last_kernel() and "persistent" in ref_asm["biped"].last_kernel()
    torch.cuda.synchronize()
    workers = [make() for _ in range(2)]          # (plans are created before the threads start)
    errors = []

    def run(asms, start_with):
        try:
            stream = torch.cuda.Stream()
            order = ["biped", "c3"] if start_with == 0 else ["c3", "biped"]
            outs = {k: tuple(torch.empty_like(t) for t in ref[k]) for k in cases}
            for i in range(200):
                k = order[i % 2]
                got = asms[k].assemble(given[k], out=outs[k], stream=stream)
                if i % 25 < 2:                         # (both plans, every 25 launches)
                    stream.synchronize()
                    for mine, theirs in zip(got, ref[k]):
                        if not torch.equal(mine, theirs):
                            errors.append("%s, launch %d: results differ" % (k, i))
            stream.synchronize()
        except Exception as exc:                       # (a failed launch: MpcasmError)
            errors.append(repr(exc))

    threads = [threading.Thread(target=run, args=(workers[i], i)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
