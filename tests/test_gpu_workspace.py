"""The persistent kernel's compact workspace (plan.py Workspace.windows): every row-set keeps only the
4-column blocks it can be non-zero in.  Same QPs as with the dense workspace and as the oracle's, on the
per-plan kernel and on the ahead-of-time one, with the descriptor path of G (small problems, incl. the
34-wide phase with its two-axis pieces) and the row-record path (wide ones)."""
import numpy as np
import pytest

from helpers import RTOL_TIGHT, assert_close
from mpcasm import problems
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


def cases(api):
    for samples, times in ((8, (6, 14)), (8, (7, 15)), (12, (10, 22)), (12, (11, 23))):
        form = problems.biped(api, problems.BipedConfig(step_samples=samples))
        form.update(step_times=np.array(times), step_count=0)
        yield "biped N=%d, %d wide" % (2 * samples, form.optim_len), form
    yield "lipm3d N=32", problems.lipm3d(api, N=32)


@pytest.mark.parametrize("jit", [1, 2], ids=["per-plan kernel", "ahead-of-time kernel"])
@pytest.mark.parametrize("on_chip", [False, True], ids=["S, U from memory", "S, U on chip"])
def test_compact_workspace_gives_the_same_qps(gpu_api, jit, on_chip):
    import torch

    from mpcasm import capi
    from mpcasm.engine import Assembler

    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        batch = 67
        for name, form in cases(gpu_api):
            lti = ["LIP"] if on_chip else []
            dense = Assembler(form, batch=batch, lti=lti, workspace="dense")
            compact = Assembler(form, batch=batch, lti=lti, workspace="compact")
            assert dense.plan.workspace.compact == 0 and compact.plan.workspace.compact == 1, name
            assert compact.plan.workspace.doubles < dense.plan.workspace.doubles
            given = np.random.default_rng(5).normal(0, 0.1, [batch, form.given_len])
            g = torch.as_tensor(given, device="cuda")
            ref = [t.cpu().numpy() for t in dense.assemble(g)]
            if "persistent" not in dense.last_kernel():
                continue                       # (C3 with S, U from memory: too big an image, staged)
            got = [t.cpu().numpy() for t in compact.assemble(g)]
            assert ("compiled for the plan" if jit == 1 else "ahead of time") in compact.last_kernel(), name
            for key, a, b in zip("PqGh", got, ref):
                assert_close(a, b, 1e-13, "%s %s" % (name, key))
            for b_ in (0, batch - 1):
                A, hh, Q, qq = orc.assemble(form, given[b_].reshape(-1, 1))
                for key, a, r in zip("PqGh", got, (Q, qq.ravel(), A, hh.ravel())):
                    assert_close(a[b_], r, RTOL_TIGHT, "%s %s" % (name, key))
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)


def test_assembler_picks_the_workspace_that_fits_two_workgroups(gpu_api):
    """``workspace="auto"`` (engine.plan_for_device): compact when the dense workspace leaves room for one
    workgroup per CU and the compact one for two -- judged with P in LDS for an assembler sized for
    streaming launches; results as the oracle's either way."""
    import torch

    from mpcasm.engine import Assembler

    n24 = problems.biped(gpu_api, problems.BipedConfig(step_samples=12))
    n24.update(step_times=np.array([10, 22]), step_count=0)
    for lti, batch, compact in (([], 64, 1), (["LIP"], 64, 0), (["LIP"], 16384, 1)):
        asm = Assembler(n24, batch=batch, lti=lti)
        assert asm.plan.workspace.compact == compact, (lti, batch)
        given = np.random.default_rng(6).normal(0, 0.1, [batch, n24.given_len])
        got = [t[:3].cpu().numpy() for t in asm.assemble(torch.as_tensor(given, device="cuda"))]
        A, hh, Q, qq = orc.assemble(n24, given[2].reshape(-1, 1))
        for key, a, r in zip("PqGh", got, (Q, qq.ravel(), A, hh.ravel())):
            assert_close(a[2], r, RTOL_TIGHT, key)
        del asm


def test_compact_workspace_partial_launches_and_halves(gpu_api):
    """``count`` < batch and the cost-only / constraints-only launches on a compact plan: the same numbers
    as the whole launch, the rest of the buffers untouched."""
    import torch

    from mpcasm.engine import Assembler

    form = problems.lipm3d(gpu_api, N=32)
    batch = 40
    asm = Assembler(form, batch=batch, lti=["LIP"])
    assert asm.plan.workspace.compact == 1
    g = torch.as_tensor(np.random.default_rng(8).normal(0, 0.1, [batch, form.given_len]), device="cuda")
    ref = [t.clone() for t in asm.assemble(g)]
    out = tuple(torch.full_like(t, -7.0) for t in ref)
    asm.assemble(g, out=out, count=13)
    for a, r in zip(out, ref):
        assert torch.equal(a[:13], r[:13]) and bool((a[13:] == -7.0).all())
    P, q, G, h = asm.assemble(g, out=tuple(torch.full_like(t, -7.0) for t in ref), want_constraints=False)
    assert G is None and h is None and torch.equal(P, ref[0]) and torch.equal(q, ref[1])
    P, q, G, h = asm.assemble(g, out=tuple(torch.full_like(t, -7.0) for t in ref), want_cost=False)
    assert P is None and q is None and torch.equal(G, ref[2]) and torch.equal(h, ref[3])
