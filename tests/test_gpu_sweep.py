"""GPU: the sweep kernel (csrc/sweep.hip) -- a dynamics compiled as ``ltv``, its own (A_k, B_k) at every
step and in every instance (BASELINE config C5), assembled without a horizon matrix -- against the
oracle on the dense S, U of ``extend_matrices_ltv``, against the fill + assembly route, and, pinned to
the reference, against the ``lti`` path where all steps share one pair."""
import numpy as np
import pytest

from helpers import RTOL, RTOL_TIGHT, assert_close, lti_tracking_problem
from mpcasm import problems
from mpcasm.plan import _H
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(x, ref):
    scale = ref.abs().max()
    err = (x - ref).abs().max()
    return float(err / scale) if float(scale) > 0 else float(err)


@pytest.fixture
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def _ltv_batch(api, B, N, rng):
    """Per-step LIPM systems (problems.ltv_lipm_steps): eight true ones, the others scaled copies --
    every instance differs."""
    thetas = rng.uniform(0, 2 * np.pi, 8)
    first = [problems.ltv_lipm_steps(api, N=N, theta=float(t)) for t in thetas]
    A = np.stack([first[i % 8][0] * (1.0 - 1e-3 * (i // 8) / max(B // 8, 1)) for i in range(B)])
    Bm = np.stack([first[i % 8][1] * (1.0 + 1e-3 * (i // 8) / max(B // 8, 1)) for i in range(B)])
    return A, Bm


def _oracle_on(form, name, extend, given_row):
    dyn = form.dynamics[name]
    saved = list(dyn.matrices)
    try:
        So, Uo = extend()
        dyn.matrices = list(Uo) + [So]
        dyn.update_definitions()
        return orc.assemble(form, given_row.reshape(-1, 1))
    finally:
        dyn.matrices = saved
        dyn.update_definitions()


def test_c5_assembly_at_2048_instances(gpu_api, torch_gpu):
    """The C5-shaped assembly (problems.lipm_ltv: N = 100, two axes, 200 unknowns, 404 lines) at the
    per-GPU batch of 2048, per-instance per-step systems and weights: every element written, P
    symmetric, q and h affine in `given`, P affine in a weight, sampled instances against the oracle."""
    torch = torch_gpu
    from mpcasm import engine

    B, N = 2048, 100
    rng = np.random.default_rng(20263)
    form = problems.lipm_ltv(gpu_api, N=N)
    A, Bm = _ltv_batch(gpu_api, B, N, rng)
    asm = engine.Assembler(form, batch=B, ltv=["LIP"])
    assert asm.plan.itab[_H["SW_OK"]] == 1
    asm.bind_ltv("LIP", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
    w = rng.uniform(0.005, 0.02, [B, 1, 1])
    asm.set_param("cost", "velocity", "weight", w)
    g0 = torch.as_tensor(rng.normal(0, 0.05, [B, form.given_len]), device="cuda")
    g1 = torch.as_tensor(rng.normal(0, 0.05, [B, form.given_len]), device="cuda")
    out = tuple(torch.full_like(t, float("nan")) for t in asm.assemble(g0))
    P, q, G, h = (t.clone() for t in asm.assemble(g0, out=out))
    assert "sweep" in asm.last_kernel(), asm.last_kernel()
    assert P.shape == (B, 2 * N, 2 * N) and G.shape == (B, 4 * N + 4, 2 * N)
    assert not any(torch.isnan(t).any().item() for t in (P, q, G, h))
    assert _rel(P.transpose(1, 2), P) <= RTOL_TIGHT
    goal = form.goals["velocity"]
    w0 = goal.weight
    try:
        for b in (0, 1, 255, 256, 1023, 1500, B - 2, B - 1):
            goal.update(weight=float(w[b, 0, 0]))
            Ao, ho, Qo, qo = _oracle_on(form, "LIP", lambda: orc.extend_matrices_ltv(N, A[b], Bm[b]),
                                        g0[b].cpu().numpy())
            assert_close(P[b].cpu().numpy(), Qo, RTOL), assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL)
            assert_close(G[b].cpu().numpy(), Ao, RTOL), assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL)
    finally:
        goal.update(weight=w0)
    # q, h affine in given; P, G independent of it
    _, q1, _, h1 = (t.clone() for t in asm.assemble(g1))
    _, qz, _, hz = (t.clone() for t in asm.assemble(torch.zeros_like(g0)))
    Ps, qs, Gs, hs = asm.assemble(g0 + g1)
    assert torch.equal(Ps, P) and torch.equal(Gs, G)
    assert _rel(q + q1, qs + qz) <= 1e-12 and _rel(h + h1, hs + hz) <= 1e-12
    # one half at a time: the same numbers
    P2, q2, _, _ = asm.assemble(g0, want_constraints=False)
    assert torch.equal(P2, P) and torch.equal(q2, q)
    out = tuple(torch.full_like(t, float("nan")) for t in (P, q, G, h))
    _, _, G2, h2 = asm.assemble(g0, out=out, want_cost=False)
    assert torch.equal(G2, G) and torch.equal(h2, h)


def test_all_steps_one_pair_is_the_lti_path(gpu_api, torch_gpu):
    """Where the reference pins this path: A_k = A, B_k = B at every step.  The sweep kernel against the
    same formulation compiled with lti= (the tiled kernel on generated tables) and against the oracle on
    the reference's own extend_matrices."""
    torch = torch_gpu
    from mpcasm import engine

    B, N = 33, 100
    rng = np.random.default_rng(7)
    form = problems.lipm_ltv(gpu_api, N=N)
    get_A, get_B, _ = gpu_api.tools.get_system_matrices("dP->CCC")
    omegas = rng.uniform(3.0, 3.6, B)
    A = np.stack([np.asarray(get_A(tau=0.1, omega=float(o)), dtype=float) for o in omegas])
    Bm = np.stack([np.asarray(get_B(tau=0.1, omega=float(o)), dtype=float).reshape(3, 1) for o in omegas])
    given = torch.as_tensor(rng.normal(0, 0.05, [B, form.given_len]), device="cuda")
    ltv = engine.Assembler(form, batch=B, ltv=["LIP"])
    ltv.bind_ltv("LIP", torch.as_tensor(np.repeat(A[:, None], N, axis=1), device="cuda"),
                 torch.as_tensor(np.repeat(Bm[:, None], N, axis=1), device="cuda"))
    lti = engine.Assembler(form, batch=B, lti=["LIP"])
    lti.bind_lti("LIP", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
    mine, theirs = ltv.assemble(given), lti.assemble(given)
    assert "sweep" in ltv.last_kernel() and "sweep" not in lti.last_kernel()
    for x, y in zip(mine, theirs):
        assert _rel(x, y) <= 1e-12
    for b in (0, B - 1):
        Ao, ho, Qo, qo = _oracle_on(form, "LIP", lambda: orc.extend_matrices(N, A[b], Bm[b]),
                                    given[b].cpu().numpy())
        Pb, qb, Gb, hb = (t[b].cpu().numpy() for t in mine)
        assert_close(Pb, Qo, RTOL_TIGHT), assert_close(qb, qo.ravel(), RTOL_TIGHT)
        assert_close(Gb, Ao, RTOL_TIGHT), assert_close(hb, ho.ravel(), RTOL_TIGHT)
    # the nominal pair of the formulation itself, before any bind_ltv: the drop-in's own numbers
    fresh = engine.Assembler(form, batch=2, ltv=["LIP"])
    Ao, ho, Qo, qo = orc.assemble(form, given[0].cpu().numpy().reshape(-1, 1))
    Pn, qn, Gn, hn = (t[0].cpu().numpy() for t in fresh.assemble(given[:2]))
    assert_close(Pn, Qo, RTOL_TIGHT), assert_close(Gn, Ao, RTOL_TIGHT)
    assert_close(qn, qo.ravel(), RTOL_TIGHT), assert_close(hn, ho.ravel(), RTOL_TIGHT)


@pytest.mark.parametrize("nx,nu,N,kw", [
    (4, 2, 20, dict(scaled=True, two_axis_limit=True)),     # combinations of states, a limit over two of them
    (3, 4, 16, dict(scheduled_cost=True)),                  # more inputs than states; a cost on part of the horizon
    (2, 1, 7, dict()),                                      # the smallest: seven unknowns
    (4, 4, 70, dict(scaled=True)),                          # 280 unknowns: two columns per thread
], ids=["outputs", "wide-input", "tiny", "two-columns"])
def test_random_per_step_systems(gpu_api, torch_gpu, nx, nu, N, kw):
    """Random per-step systems of every size the kernel takes, against the fill + staged-assembly route
    (mpcasm_fill_su(ltv) -> S, U bound as sources) and the oracle."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(100 * nx + N)
    form, _, _ = lti_tracking_problem(gpu_api, rng, nx, nu, N, **kw)
    B = 9
    A = np.stack([np.stack([problems.random_lti_matrices(rng, nx, nu)[0] for _ in range(N)]) for _ in range(B)])
    Bm = np.stack([np.stack([problems.random_lti_matrices(rng, nx, nu)[1] for _ in range(N)]) for _ in range(B)])
    At, Bt = torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda")
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    asm = engine.Assembler(form, batch=B, ltv=["plant"])
    asm.bind_ltv("plant", At, Bt)
    out = tuple(torch.full_like(t, float("nan")) for t in asm.assemble(given))
    mine = tuple(t.clone() for t in asm.assemble(given, out=out))
    assert "sweep" in asm.last_kernel()
    assert not any(torch.isnan(t).any().item() for t in mine)
    ref = engine.Assembler(form, batch=B)
    ref.set_option(capi.OPT_PATH, 2)
    S, U = engine.fill_su(At, Bt, N, ltv=True)
    for j in range(nu):
        ref.bind_source(("plant", j), U[:, j])
    ref.bind_source(("plant", nu), S)
    for x, y in zip(mine, ref.assemble(given)):
        assert _rel(x, y) <= 1e-12
    for b in (0, B - 1):
        Ao, ho, Qo, qo = _oracle_on(form, "plant", lambda: orc.extend_matrices_ltv(N, A[b], Bm[b]),
                                    given[b].cpu().numpy())
        Pb, qb, Gb, hb = (t[b].cpu().numpy() for t in mine)
        assert_close(Pb, Qo, RTOL_TIGHT), assert_close(qb, qo.ravel(), RTOL_TIGHT)
        assert_close(Gb, Ao, RTOL_TIGHT), assert_close(hb, ho.ravel(), RTOL_TIGHT)


def test_arrows_of_their_own_per_line_and_many_lines_per_step(gpu_api, torch_gpu):
    """A limit whose arrow changes from line to line (the weights of G's lines are then kept per line,
    not per limit) and more lines at one step than the kernel fetches ahead (twelve limits on every
    step), against the oracle."""
    torch = torch_gpu
    from mpcasm import engine

    api = gpu_api
    rng = np.random.default_rng(9)
    nx, nu, N, B = 3, 2, 24, 5
    A0, B0 = problems.random_lti_matrices(rng, nx, nu)
    system = api.ControlSystem(["u0", "u1"], ["s0", "s1", "s2"], A0, B0, axes=["_x", "_y"])
    ext = api.ExtendedSystem.from_cotrol_system(system, "x", N)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    form.incorporate_goal("track", api.Cost("s0", 0.7, aim=[0.3, -0.2], axes=["_x", "_y"]))
    form.incorporate_goal("effort", api.Cost("u1", 0.2, axes=["_x", "_y"]))
    limits = [api.Constraint("s%d" % (k % 3), 2.0 + k, axes=["_x", "_y"], arrow=[0.6 + 0.1 * k, 0.8 - 0.1 * k])
              for k in range(12)]
    limits.append(api.Constraint("s1", 3.0, axes=["_x", "_y"], arrow=rng.standard_normal((N, 2))))
    form.incorporate_constraint("many", limits)
    form.identify_qp_domain(["u0_x", "u1_x", "u0_y", "u1_y"])
    form.make_preview_matrices()
    A = np.stack([np.stack([problems.random_lti_matrices(rng, nx, nu)[0] for _ in range(N)]) for _ in range(B)])
    Bm = np.stack([np.stack([problems.random_lti_matrices(rng, nx, nu)[1] for _ in range(N)]) for _ in range(B)])
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    for drop_last in (False, True):           # (with and without the limit that has per-line arrows)
        use = limits if not drop_last else limits[:-1]
        asm = engine.Assembler(form, batch=B, ltv=["plant"], limits=use)
        asm.bind_ltv("plant", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
        out = tuple(torch.full_like(t, float("nan")) for t in asm.assemble(given))
        mine = tuple(t.clone() for t in asm.assemble(given, out=out))
        assert "sweep" in asm.last_kernel() and not any(torch.isnan(t).any().item() for t in mine)
        for b in (0, B - 1):
            dyn = form.dynamics["plant"]
            saved = list(dyn.matrices)
            try:
                So, Uo = orc.extend_matrices_ltv(N, A[b], Bm[b])
                dyn.matrices = list(Uo) + [So]
                dyn.update_definitions()
                PM = orc.preview_matrices(form)
                g = given[b].cpu().numpy().reshape(-1, 1)
                parts = [orc.qp_constraint(PM, l, g) for l in use]
                Ao, ho = np.vstack([x[0] for x in parts]), np.vstack([x[1] for x in parts])
                Qo, qo = orc.qp_all_costs(form, PM, g)
            finally:
                dyn.matrices = saved
                dyn.update_definitions()
            Pb, qb, Gb, hb = (t[b].cpu().numpy() for t in mine)
            assert_close(Pb, Qo, RTOL_TIGHT), assert_close(qb, qo.ravel(), RTOL_TIGHT)
            assert_close(Gb, Ao, RTOL_TIGHT), assert_close(hb, ho.ravel(), RTOL_TIGHT)


def test_what_the_sweep_kernel_does_not_take(gpu_api, torch_gpu):
    """ValueError when the formulation cannot be assembled step by step; the preview entry points have no
    horizon tables to read for such a plan."""
    from mpcasm import capi, engine

    rng = np.random.default_rng(2)
    form, _, _ = lti_tracking_problem(gpu_api, rng, 3, 3, 16, extra_unknown=True)
    with pytest.raises(ValueError):
        engine.Assembler(form, batch=2, ltv=["plant"])
    form, _, _ = lti_tracking_problem(gpu_api, rng, 3, 3, 16)
    asm = engine.Assembler(form, batch=2, ltv=["plant"])
    with pytest.raises(ValueError):
        asm.bind_source(("plant", 0), np.zeros((16, 16, 3)))
    with pytest.raises(capi.MpcasmError):
        asm.preview_rows(np.zeros((2, form.given_len)), np.zeros((2, form.optim_len)))
