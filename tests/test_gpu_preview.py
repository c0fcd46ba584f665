"""GPU: f2 on the device without a preview matrix in memory -- mpcasm_preview_direct and
mpcasm_goal_distance (body.py:209-234) for whole batches, against the oracle's dense
preview matrices and numpy, and through linearity over the batch."""
import numpy as np
import pytest

from helpers import RTOL_TIGHT, assert_close
from mpcasm import problems
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


def _biped(api, N):
    conf = problems.BipedConfig(step_samples=N // 2)
    form = problems.biped(api, conf)
    form.update(step_times=np.array([N // 2 - 2, N - 2]), step_count=0)
    return form


def _dense_rows(form, plan, given, optim):
    """All preview rows of one instance from the oracle's dense [Mg | Mo]."""
    PM = orc.preview_matrices(form)
    out = np.zeros(plan.pmrows)
    for var, (r0, rows) in plan.pm_rows.items():
        out[r0:r0 + rows] = (PM[var][0] @ given + PM[var][1] @ optim).ravel()
    return out


@pytest.mark.parametrize("lti", [False, True])
def test_batched_preview_and_goal_distances(gpu_api, lti):
    """B = 4096 walkers of the C2 biped with their own given, solver answer and velocity aim:
    every row of every definition and every goal's squared distance, sampled instances against
    the oracle, all of them through linearity in (given, optim)."""
    import torch

    from mpcasm import engine

    form = _biped(gpu_api, 16)
    B = 4096
    rng = np.random.default_rng(3)
    asm = engine.Assembler(form, batch=B, lti=["LIP"] if lti else ())
    given = rng.normal(0, 0.1, [B, form.given_len])
    optim = rng.normal(0, 0.5, [B, form.optim_len])
    aims = rng.uniform(0, 0.6, [B, 1, 1])
    vel = "track vel_x"
    asm.set_param("cost", vel, "aim", aims)
    gt, xt = torch.as_tensor(given, device="cuda"), torch.as_tensor(optim, device="cuda")
    rows = asm.preview_rows(gt, xt)
    assert rows.shape == (B, asm.plan.pmrows)
    dist = asm.goal_distance(form, rows)
    names = asm.goal_terms(form)[1]
    assert dist.shape == (B, len(names)) and names == list(form.goals.keys())
    # the same distances with the rows never leaving the chip (mpcasm_preview_goal_distance): the whole
    # batch against the two-step path, an odd count (the last block of four is not full), a caller's buffer
    fused = asm.full_goal_distances(form, gt, xt)
    assert float((fused - dist).abs().max() / dist.abs().max()) <= 1e-13
    part = torch.full((B, len(names)), float("nan"), dtype=torch.float64, device="cuda")
    asm.full_goal_distances(form, gt, xt, out=part, count=B - 3)
    assert torch.equal(part[:B - 3], fused[:B - 3]) and torch.isnan(part[B - 3:]).all()
    R, D = rows.cpu().numpy(), dist.cpu().numpy()
    goal = form.goals[vel]
    saved = np.array(goal.aim, copy=True)
    try:
        for b in (0, 1, 2047, B - 1):
            ref = _dense_rows(form, asm.plan, given[b].reshape(-1, 1), optim[b].reshape(-1, 1))
            assert_close(R[b], ref, RTOL_TIGHT, "rows of instance %d" % b)
            goal.update(aim=aims[b, 0])
            for gi, name in enumerate(names):
                g = form.goals[name]
                want = 0.0
                for i, axis in enumerate(g.axes):
                    r0, n = asm.plan.pm_rows[g.variable + axis]
                    v = ref[r0:r0 + n] - np.asarray(g.aim, dtype=float)[:, i]
                    want += float(v @ v)
                assert abs(D[b, gi] - want) <= 1e-12 * max(1.0, abs(want)), (b, name)
    finally:
        goal.update(aim=saved)
    # linear in (given, optim) over the whole batch: f(a) + f(b) = f(a + b)
    g2 = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    x2 = torch.as_tensor(rng.normal(0, 0.5, [B, form.optim_len]), device="cuda")
    both = asm.preview_rows(gt + g2, xt + x2)
    parts = rows + asm.preview_rows(g2, x2)
    assert float((both - parts).abs().max() / both.abs().max()) <= 1e-13
    # ... and the rows of the old two-pass path (preview matrices in HBM, then a GEMV)
    if not lti:
        old = asm.preview(asm.preview_matrices(), gt, xt)
        assert float((old - rows).abs().max() / rows.abs().max()) <= 1e-13


def test_preview_of_a_wide_problem(gpu_api):
    """C4 shape (no = 384): 12 states and 6 inputs of 64 rows each, per-instance systems from
    (A, B); the state rows are the simulated trajectory."""
    import torch

    from mpcasm import engine

    nx, nu, N, B = 12, 6, 64, 257
    rng = np.random.default_rng(8)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    As, Bs = zip(*(problems.random_lti_matrices(rng, nx, nu) for _ in range(B)))
    A, Bm = np.stack(As), np.stack(Bs)
    asm = engine.Assembler(form, batch=B, lti=["plant"])
    asm.bind_lti("plant", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
    x0 = rng.normal(0, 0.3, [B, form.given_len])
    u = rng.normal(0, 0.5, [B, form.optim_len])
    rows = asm.preview_rows(x0, u).cpu().numpy()
    for b in (0, 128, B - 1):
        x = x0[b][form.given_ID["x0"]]
        U = np.stack([u[b][form.optim_ID["u%d" % j]] for j in range(nu)], axis=1)     # (N, nu)
        traj = []
        for k in range(N):
            x = A[b] @ x + Bm[b] @ U[k]
            traj.append(x)
        traj = np.asarray(traj)
        for i in range(nx):
            r0, n = asm.plan.pm_rows["s%d" % i]
            assert_close(rows[b, r0:r0 + n], traj[:, i], 1e-11, "state %d" % i)
        for j in range(nu):
            r0, n = asm.plan.pm_rows["u%d" % j]
            assert np.array_equal(rows[b, r0:r0 + n], U[:, j])


def test_drop_in_preview_and_distances_use_the_device(gpu_api):
    """Formulation.preview / goal_distance / full_goal_distance (B = 1 view of the same
    kernels) against the reference's formulas on the oracle's matrices."""
    form = problems.body_case(gpu_api)
    rng = np.random.default_rng(1)
    given = rng.standard_normal([form.given_len, 1])
    optim = rng.standard_normal([form.optim_len, 1])
    PM = orc.preview_matrices(form)
    total = 0.0
    for name, goal in form.goals.items():
        want = 0.0
        for i, axis in enumerate(goal.axes):
            v = orc.preview(PM, given, optim, goal.variable + axis) - np.asarray(goal.aim)[:, i]
            want += float(v.T @ v)
        assert abs(form.goal_distance(given, optim, name) - want) <= 1e-12 * max(1.0, want)
        total += want
    assert abs(form.full_goal_distance(given, optim) - total) <= 1e-12 * max(1.0, total)
    with pytest.raises(KeyError):
        form.goal_distance(given, optim, "no such goal")
