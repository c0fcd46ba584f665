"""GPU: the tiled kernel (csrc/tiled.hip) -- wide problems (no >= 128) without a workspace in
HBM -- against the staged pipeline and the oracle; horizon tables generated from per-instance
(A, B) against the K1 fill; C4 at its per-GPU batch of 8192 in ONE call."""
import numpy as np
import pytest

from helpers import RTOL, RTOL_TIGHT, assert_close, lti_tracking_problem
from mpcasm import problems
from mpcasm.plan import _H
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(x, ref):
    import torch

    scale = ref.abs().max()
    err = (x - ref).abs().max()
    return float(err / scale) if float(scale) > 0 else float(err)


@pytest.fixture
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.mark.parametrize("nx,nu,N,B,seed", [
    (5, 3, 48, 37, 1),       # no = 144: the second column block is mostly padding
    (4, 6, 40, 16, 2),       # no = 240: stages of 16, 16 and 8 rows
    (12, 6, 64, 24, 20262),  # the C4 shape
    (3, 2, 100, 9, 5),       # no = 200, N = 100: seven stages per term, the last of 4 rows
])
def test_tiled_against_staged_and_oracle(gpu_api, torch_gpu, nx, nu, N, B, seed):
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(seed)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    w = rng.uniform(0.1, 1.0, [B, 1, 1])
    til = engine.Assembler(form, batch=B)
    assert til.plan.itab[_H["T_OK"]] == 1                   # this plan runs on the tiled kernel
    til.set_param("cost", "track s0", "weight", w)
    # garbage in the result buffers first: every element must be written, zeros included
    out = tuple(torch.full_like(t, float("nan")) for t in til.assemble(given))
    Pt, qt, Gt, ht = (t.clone() for t in til.assemble(given, out=out))
    assert not any(torch.isnan(t).any().item() for t in (Pt, qt, Gt, ht))
    ref = engine.Assembler(form, batch=B)
    ref.set_option(capi.OPT_PATH, 2)                          # the staged pipeline
    ref.set_param("cost", "track s0", "weight", w)
    Ps, qs, Gs, hs = ref.assemble(given)
    assert max(_rel(Pt, Ps), _rel(qt, qs), _rel(Gt, Gs), _rel(ht, hs)) <= RTOL_TIGHT
    goal = form.goals["track s0"]
    w0 = goal.weight
    try:
        for b in (0, B - 1):
            goal.update(weight=float(w[b, 0, 0]))
            Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
            assert_close(Pt[b].cpu().numpy(), Qo, RTOL_TIGHT), assert_close(qt[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
            assert_close(Gt[b].cpu().numpy(), Ao, RTOL_TIGHT), assert_close(ht[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)
    finally:
        goal.update(weight=w0)
    # one half at a time: the same numbers
    P2, q2, _, _ = til.assemble(given, want_constraints=False)
    assert torch.equal(P2, Pt) and torch.equal(q2, qt)
    out = tuple(torch.full_like(t, float("nan")) for t in (Pt, qt, Gt, ht))
    _, _, G2, h2 = til.assemble(given, out=out, want_cost=False)
    assert torch.equal(G2, Gt) and torch.equal(h2, ht)


@pytest.mark.parametrize("nx,nu,N,B,kw", [
    (5, 3, 48, 67, {}),                                              # no = 144; a ragged last group of instances
    (12, 6, 64, 40, {}),                                             # the C4 shape
    (4, 6, 40, 33, dict(scaled=True, two_axis_limit=True)),          # a derived variable; a row of G over two rows of V
    (3, 4, 40, 300, dict(extra_unknown=True, scheduled_cost=True)),  # a diagonal term with a weight of its own; a cost on part of the horizon
    (3, 3, 100, 35, dict(given_input=True)),                         # d = Mg given from a given input (200 unknowns)
])
def test_one_model_for_the_whole_batch_is_the_shared_form(gpu_api, torch_gpu, nx, nu, N, B, kw):
    """Every source shared by the batch (a fleet on one model): P_b = sum_g w_b[g] K_g with the K_g of the
    general kernel on unit weights, rows of G as scaled copies of the rows composed once, q and h from
    them -- with EVERY parameter an instance's own (weights, aims, arrows, centers, extremes), against
    the general kernel on the same launch (MPCASM_OPT_PATH 3), against the oracle for sampled instances,
    one half at a time, and with NaNs in the result buffers first."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(nx * 100 + nu)
    form, _, _ = lti_tracking_problem(gpu_api, rng, nx, nu, N, **kw)
    asm = engine.Assembler(form, batch=B)
    assert asm.plan.itab[_H["T_OK"]] == 1
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    # every field of every cost and limit differs from instance to instance
    host = asm.params.cpu().numpy().copy()
    for (kind, name, field), (start, rows, cols) in asm.plan.param_slots.items():
        block = host[:, start:start + rows * cols]
        if field == "weight":
            block *= rng.uniform(0.5, 2.0, [B, 1])
        elif field == "arrow":
            block *= rng.uniform(0.5, 1.5, [B, 1])     # (no sign change: Constraint.update would renormalise the host object)
        else:
            block += rng.normal(0, 0.2, block.shape)
    asm.params.copy_(torch.as_tensor(host, device="cuda"))
    out = tuple(torch.full_like(t, float("nan")) for t in asm.assemble(given))
    Ps, qs, Gs, hs = (t.clone() for t in asm.assemble(given, out=out))
    assert "shared" in asm.last_kernel(), asm.last_kernel()
    assert not any(torch.isnan(t).any().item() for t in (Ps, qs, Gs, hs))
    asm.set_option(capi.OPT_PATH, 3)
    Pg, qg, Gg, hg = (t.clone() for t in asm.assemble(given))
    assert asm.last_kernel() == "tiled_assemble_kernel", asm.last_kernel()
    assert max(_rel(Ps, Pg), _rel(qs, qg), _rel(Gs, Gg), _rel(hs, hg)) <= RTOL_TIGHT
    asm.set_option(capi.OPT_PATH, -1)
    P2, q2, _, _ = asm.assemble(given, want_constraints=False)
    assert torch.equal(P2, Ps) and torch.equal(q2, qs)
    out = tuple(torch.full_like(t, float("nan")) for t in (Ps, qs, Gs, hs))
    _, _, G2, h2 = asm.assemble(given, out=out, want_cost=False)
    assert torch.equal(G2, Gs) and torch.equal(h2, hs)
    # the oracle, with the instance's own numbers in the formulation's objects
    limits = orc.all_limits(form)
    saved = {(k, n, f): np.array(getattr(form.goals[n] if k == "cost" else limits[n], f))
             for (k, n, f) in asm.plan.param_slots}
    try:
        for b in (0, B // 2, B - 1):
            for (kind, name, field), (start, rows, cols) in asm.plan.param_slots.items():
                obj = form.goals[name] if kind == "cost" else limits[name]
                value = host[b, start:start + rows * cols].reshape(rows, cols)
                obj.update(**{field: float(value[0, 0]) if field == "weight" else value})
            Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
            assert_close(Ps[b].cpu().numpy(), Qo, RTOL_TIGHT), assert_close(qs[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
            assert_close(Gs[b].cpu().numpy(), Ao, RTOL_TIGHT), assert_close(hs[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)
    finally:
        for (kind, name, field), value in saved.items():
            obj = form.goals[name] if kind == "cost" else limits[name]
            obj.update(**{field: float(value.ravel()[0]) if field == "weight" else value})


@pytest.mark.parametrize("path", [0, 1, 4, 3], ids=["scan", "scan with pre-passes", "toeplitz", "general"])
@pytest.mark.parametrize("nx,nu,N,B,seed", [(5, 3, 48, 19, 11), (12, 6, 64, 24, 12), (3, 4, 40, 9, 13),
                                            (3, 2, 100, 7, 14)])
def test_horizon_tables_from_the_systems_own_matrices(gpu_api, torch_gpu, nx, nu, N, B, seed, path):
    """Plans compiled with lti=[...] on the tiled kernel: a pre-pass builds S and the compact
    Toeplitz tables of U from per-instance (A, B) by the reference's recurrence (tools.py:24-29).
    All three forms of the kernel -- P summed along diagonals (every cost is the full horizon of a
    state: the scan form, horizons up to 64), operands read out of the table in LDS (every cost row is
    a window of it: MPCASM_OPT_PATH 4), and the general form that composes tiles (MPCASM_OPT_PATH 3)
    -- against the staged pipeline fed the K1 fill's S, U of the same systems, and the oracle."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(seed)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    As, Bs = zip(*(problems.random_lti_matrices(rng, nx, nu) for _ in range(B)))
    A, Bm = np.stack(As), np.stack(Bs)
    At, Bt = torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda")
    lti = engine.Assembler(form, batch=B, lti=["plant"])
    assert lti.plan.itab[_H["T_TOEPLITZ"]] == 1
    lti.set_option(capi.OPT_PATH, path)
    lti.bind_lti("plant", At, Bt)
    out = tuple(torch.full_like(t, float("nan")) for t in lti.assemble(given))
    Pl, ql, Gl, hl = (t.clone() for t in lti.assemble(given, out=out))
    assert not any(torch.isnan(t).any().item() for t in (Pl, ql, Gl, hl))
    assert "tiled" in lti.last_kernel()
    assert ("scan" in lti.last_kernel()) == (path in (0, 1) and N <= 64), lti.last_kernel()
    assert lti.plan.itab[_H["T_SCAN_FUSED"]] == (1 if N <= 64 else 0)      # (path 0: the kernel makes its own table and d; 1: pre-passes)
    # one half at a time: the same numbers
    P2, q2, _, _ = lti.assemble(given, want_constraints=False)
    assert torch.equal(P2, Pl) and torch.equal(q2, ql)
    out = tuple(torch.full_like(t, float("nan")) for t in (Pl, ql, Gl, hl))
    _, _, G2, h2 = lti.assemble(given, out=out, want_cost=False)
    assert torch.equal(G2, Gl) and torch.equal(h2, hl)
    ref = engine.Assembler(form, batch=B)
    ref.set_option(capi.OPT_PATH, 2)
    S, U = engine.fill_su(At, Bt, N)
    for j in range(nu):
        ref.bind_source(("plant", j), U[:, j])
    ref.bind_source(("plant", nu), S)
    Ps, qs, Gs, hs = ref.assemble(given)
    assert max(_rel(Pl, Ps), _rel(ql, qs), _rel(Gl, Gs), _rel(hl, hs)) <= 1e-12
    dyn = form.dynamics["plant"]
    saved = list(dyn.matrices)
    try:
        for b in (0, B - 1):
            So, Uo = orc.extend_matrices(N, A[b], Bm[b])
            dyn.matrices = Uo + [So]
            dyn.update_definitions()
            Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
            assert_close(Pl[b].cpu().numpy(), Qo, RTOL), assert_close(ql[b].cpu().numpy(), qo.ravel(), RTOL)
            assert_close(Gl[b].cpu().numpy(), Ao, RTOL), assert_close(hl[b].cpu().numpy(), ho.ravel(), RTOL)
    finally:
        dyn.matrices = saved
        dyn.update_definitions()


@pytest.mark.parametrize("nx,nu,N,kw", [
    (4, 6, 40, dict(scaled=True)),                                   # a cost on 2.5 x a state
    (3, 4, 40, dict(extra_unknown=True, scaled=True)),               # unknowns that are no input of the plant
    (4, 5, 32, dict(given_input=True, two_axis_limit=True)),         # a GIVEN input; rows of G over two states
    (6, 8, 16, dict()),                                              # eight column blocks of 16
    (16, 4, 64, dict()),                                             # sixteen terms
    (4, 4, 40, dict(scheduled_cost=True)),                           # part of the horizon: NO scan form
], ids=["scaled", "slack", "given-input", "8-blocks", "16-terms", "scheduled"])
def test_scan_form_against_the_toeplitz_form_and_the_oracle(gpu_api, torch_gpu, nx, nu, N, kw):
    """The scan form (P summed along diagonals) on what tells it apart from the matrix-core forms:
    coefficients, unknowns outside the plant's inputs (their rows and columns of P hold the diagonal
    terms only), a given input (no column block), rows of G that are no single state row (composed
    behind the rest), every instantiation's limits -- against the Toeplitz form (MPCASM_OPT_PATH 4)
    on the same plan and against the oracle with every instance's own system."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(1000 * nx + N)
    form, _, _ = lti_tracking_problem(gpu_api, rng, nx, nu, N, **kw)
    B = 11
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    As, Bs = zip(*(problems.random_lti_matrices(rng, nx, nu) for _ in range(B)))
    A, Bm = np.stack(As), np.stack(Bs)
    results = {}
    for path in (0, 4):
        asm = engine.Assembler(form, batch=B, lti=["plant"])
        asm.set_option(capi.OPT_PATH, path)
        asm.bind_lti("plant", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
        w = rng.uniform(0.1, 1.0, [B, 1, 1]) if path == 0 else w
        asm.set_param("cost", "track s0", "weight", w)
        out = tuple(torch.full_like(t, float("nan")) for t in asm.assemble(given))
        results[path] = tuple(t.clone() for t in asm.assemble(given, out=out))
        assert not any(torch.isnan(t).any().item() for t in results[path])
        scan = asm.plan.itab[_H["T_SCAN"]] > 0
        assert scan == (not kw.get("scheduled_cost"))
        assert ("scan" in asm.last_kernel()) == (scan and path == 0), asm.last_kernel()
        if path == 0:
            P2, q2, _, _ = asm.assemble(given, want_constraints=False)
            assert torch.equal(P2, results[0][0]) and torch.equal(q2, results[0][1])
            out = tuple(torch.full_like(t, float("nan")) for t in results[0])
            _, _, G2, h2 = asm.assemble(given, out=out, want_cost=False)
            assert torch.equal(G2, results[0][2]) and torch.equal(h2, results[0][3])
    for mine, theirs in zip(results[0], results[4]):
        assert _rel(mine, theirs) <= RTOL_TIGHT
    dyn, goal = form.dynamics["plant"], form.goals["track s0"]
    saved, w0 = list(dyn.matrices), goal.weight
    try:
        for b in (0, 5, B - 1):
            So, Uo = orc.extend_matrices(N, A[b], Bm[b])
            dyn.matrices = Uo + [So]
            dyn.update_definitions()
            goal.update(weight=float(w[b, 0, 0]))
            Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
            Pl, ql, Gl, hl = (t[b].cpu().numpy() for t in results[0])
            assert_close(Pl, Qo, RTOL_TIGHT), assert_close(ql, qo.ravel(), RTOL_TIGHT)
            assert_close(Gl, Ao, RTOL_TIGHT), assert_close(hl, ho.ravel(), RTOL_TIGHT)
    finally:
        dyn.matrices = saved
        dyn.update_definitions()
        goal.update(weight=w0)


def test_crossed_cost_and_two_axis_constraint_on_the_tiled_kernel(gpu_api, torch_gpu):
    """What the random LTI problems do not have: a crossed cost (A != B rows, non-symmetric P:
    every block pair is computed), a derived variable (rows of several entries: the general
    compose path), a constraint over two axes (rows of G that ride on no stage) and an L."""
    torch = torch_gpu
    from mpcasm import capi, engine

    api = gpu_api
    rng = np.random.default_rng(77)
    N, nx, nu = 32, 4, 5
    A, Bm = problems.random_lti_matrices(rng, nx, nu)
    axes = ["_x", "_y"]
    inputs = ["u%d" % j for j in range(nu)]
    states = ["s%d" % i for i in range(nx)]
    system = api.ControlSystem(inputs, states, A, Bm, axes=axes)
    ext = api.ExtendedSystem.from_cotrol_system(system, "x", N)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    for ax in axes:
        form.incorporate_definition("mix" + ax, api.LineCombo(
            {"s0" + ax: 1.0, "s1" + ax: np.diag(rng.standard_normal(N)), "u0" + ax: 0.5}))
    form.incorporate_goal("track", api.Cost("s0", 0.7, aim=[0.3, -0.2], axes=axes))
    form.incorporate_goal("crossed", api.Cost("mix", 0.4, aim=[0.1, 0.0], axes=axes,
                                              cross="s2", cross_aim=[0.0, 0.5]))
    form.incorporate_goal("effort", api.Cost("u1", 0.2, axes=axes))
    L = [rng.standard_normal((6, N)), rng.standard_normal((6, N))]
    form.incorporate_constraint("cone", [
        api.Constraint("s3", 4.0, axes=axes, arrow=[0.6, 0.8]),
        api.Constraint("mix", 2.0, axes=axes, arrow=rng.standard_normal((6, 2)), L=L),
        api.Constraint("s1", 3.0, axes=["_x"], arrow=[-1]),
    ])
    form.identify_qp_domain([u + ax for ax in axes for u in inputs])
    form.make_preview_matrices()
    assert form.optim_len == 2 * nu * N
    B = 13
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    til = engine.Assembler(form, batch=B)
    out = tuple(torch.full_like(t, float("nan")) for t in til.assemble(given))
    Pt, qt, Gt, ht = (t.clone() for t in til.assemble(given, out=out))
    assert not any(torch.isnan(t).any().item() for t in (Pt, qt, Gt, ht))
    ref = engine.Assembler(form, batch=B)
    ref.set_option(capi.OPT_PATH, 2)
    Ps, qs, Gs, hs = ref.assemble(given)
    assert max(_rel(Pt, Ps), _rel(qt, qs), _rel(Gt, Gs), _rel(ht, hs)) <= RTOL_TIGHT
    assert _rel(Pt.transpose(1, 2), Pt) > 1e-3                # the crossed cost: P is not symmetric
    for b in (0, B - 1):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
        assert_close(Pt[b].cpu().numpy(), Qo, RTOL_TIGHT), assert_close(qt[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
        assert_close(Gt[b].cpu().numpy(), Ao, RTOL_TIGHT), assert_close(ht[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)


def test_c4_at_its_per_gpu_batch_in_one_call(gpu_api, torch_gpu):
    """C4 (nx=12 nu=6 N=64: no=384, nc=1536) at 8192 instances in ONE call -- P 9.7 GB, G 38.7 GB,
    no workspace -- with a different system in every instance: P symmetric, G = [+V; -V] per state,
    q and h affine in `given`, P affine in a per-instance weight, instances against the oracle."""
    torch = torch_gpu
    from mpcasm import engine

    nx, nu, N, B = 12, 6, 64, 8192
    if torch.cuda.get_device_properties(0).total_memory < 120e9:
        pytest.skip("needs 60 GB of device memory")
    rng = np.random.default_rng(20262)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    base = [problems.random_lti_matrices(rng, nx, nu) for _ in range(64)]
    scale = 1.0 - 0.05 * rng.random(B)
    A = np.stack([base[b % 64][0] * scale[b] for b in range(B)])
    Bm = np.stack([base[b % 64][1] * (2.0 - scale[b]) for b in range(B)])
    asm = engine.Assembler(form, batch=B, lti=["plant"])
    asm.bind_lti("plant", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
    w = rng.uniform(0.1, 1.0, [B, 1, 1])
    asm.set_param("cost", "track s0", "weight", w)
    g0 = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    g1 = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    P, q, G, h = asm.assemble(g0)
    assert P.shape == (B, 384, 384) and G.shape == (B, 1536, 384)
    q0, h0 = q.clone(), h.clone()
    # symmetric, block by block (chunks of instances: no 10 GB temporaries)
    for lo in range(0, B, 512):
        blk = P[lo:lo + 512]
        assert _rel(blk.transpose(1, 2), blk) <= RTOL_TIGHT
    # every state's limits: rows [+V (64); -V (64)]
    Gv = G.view(B, nx, 2, N, 384)
    for lo in range(0, B, 256):
        assert torch.equal(Gv[lo:lo + 256, :, 0], -Gv[lo:lo + 256, :, 1])
    # eight instances against the oracle (their own systems and weights)
    dyn, goal = form.dynamics["plant"], form.goals["track s0"]
    saved, w0 = list(dyn.matrices), goal.weight
    try:
        assert "scan" in asm.last_kernel(), asm.last_kernel()
        for b in (0, 1, 255, 256, 4097, 6000, B - 2, B - 1):        # (first / last workgroups of a round)
            So, Uo = orc.extend_matrices(N, A[b], Bm[b])
            dyn.matrices = Uo + [So]
            dyn.update_definitions()
            goal.update(weight=float(w[b, 0, 0]))
            Ao, ho, Qo, qo = orc.assemble(form, g0[b].cpu().numpy().reshape(-1, 1))
            assert_close(P[b].cpu().numpy(), Qo, RTOL), assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL)
            assert_close(G[b].cpu().numpy(), Ao, RTOL), assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL)
    finally:
        dyn.matrices = saved
        dyn.update_definitions()
        goal.update(weight=w0)
    # q, h affine in given: f(g0) + f(g1) = f(g0 + g1) + f(0); P, G do not depend on it
    Pa = P[::1024].clone()
    _, q1, _, h1 = (t.clone() for t in asm.assemble(g1))
    _, qz, _, hz = (t.clone() for t in asm.assemble(torch.zeros_like(g0)))
    Pb, qs, _, hs = asm.assemble(g0 + g1)
    assert torch.equal(Pb[::1024], Pa)
    assert _rel(q0 + q1, qs + qz) <= 1e-12 and _rel(h0 + h1, hs + hz) <= 1e-12


def test_c4_with_one_system_for_the_batch_in_one_call(gpu_api, torch_gpu):
    """C4 at 8192 instances in ONE call with S, U of ONE system read from memory (the shared-model form): every
    weight an instance's own, `given` per instance.  P symmetric and exactly the weighted sum it claims to be
    (affine in each instance's weights: three launches with scaled weights), G the same for all instances
    (no arrow differs here), q and h affine in `given`, sampled instances against the oracle and the whole
    launch against the general form on a slice."""
    torch = torch_gpu
    from mpcasm import capi, engine

    nx, nu, N, B = 12, 6, 64, 8192
    if torch.cuda.get_device_properties(0).total_memory < 120e9:
        pytest.skip("needs 60 GB of device memory")
    rng = np.random.default_rng(20263)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    asm = engine.Assembler(form, batch=B)
    names = [n for n in form.goals if n.startswith("track")]
    ws = {n: rng.uniform(0.1, 1.0, [B, 1, 1]) for n in names}
    for n in names:
        asm.set_param("cost", n, "weight", ws[n])
    g0 = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    g1 = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    P, q, G, h = asm.assemble(g0)
    assert "shared" in asm.last_kernel(), asm.last_kernel()
    q0, h0 = q.clone(), h.clone()
    for lo in range(0, B, 512):
        blk = P[lo:lo + 512]
        assert _rel(blk.transpose(1, 2), blk) <= RTOL_TIGHT
    for lo in range(0, B, 256):
        assert torch.equal(G[lo:lo + 256], G[:1].expand(min(256, B - lo), -1, -1))
    goals = {n: form.goals[n] for n in names}
    saved = {n: goals[n].weight for n in names}
    try:
        for b in (0, 15, 16, 4097, B - 1):                            # (first / last instances of the workgroups' groups)
            for n in names:
                goals[n].update(weight=float(ws[n][b, 0, 0]))
            Ao, ho, Qo, qo = orc.assemble(form, g0[b].cpu().numpy().reshape(-1, 1))
            assert_close(P[b].cpu().numpy(), Qo, RTOL), assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL)
            assert_close(G[b].cpu().numpy(), Ao, RTOL), assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL)
    finally:
        for n in names:
            goals[n].update(weight=saved[n])
    # the general form on the first 40 instances of the same launch
    small = engine.Assembler(form, batch=40)
    for n in names:
        small.set_param("cost", n, "weight", ws[n][:40])
    small.set_option(capi.OPT_PATH, 3)
    Pg, qg, Gg, hg = small.assemble(g0[:40])
    assert small.last_kernel() == "tiled_assemble_kernel"
    assert max(_rel(P[:40], Pg), _rel(q[:40], qg), _rel(G[:40], Gg), _rel(h[:40], hg)) <= RTOL_TIGHT
    # q, h affine in given; P does not depend on it
    Pa = P[::1024].clone()
    _, q1, _, h1 = (t.clone() for t in asm.assemble(g1))
    _, qz, _, hz = (t.clone() for t in asm.assemble(torch.zeros_like(g0)))
    Pb, qs, _, hs = asm.assemble(g0 + g1)
    assert torch.equal(Pb[::1024], Pa)
    assert _rel(q0 + q1, qs + qz) <= 1e-12 and _rel(h0 + h1, hs + hz) <= 1e-12


def _two_systems(api, rng, N, nx1, nu1, nx2, nu2, optim_second=True):
    """Two independent LTI plants in one formulation (inputs u*, v*), tracking + bounds on states of
    both; with ``optim_second`` False the second plant's inputs are GIVEN (they enter d, h, q only)."""
    A1, B1 = problems.random_lti_matrices(rng, nx1, nu1)
    A2, B2 = problems.random_lti_matrices(rng, nx2, nu2)
    s1 = api.ControlSystem(["u%d" % j for j in range(nu1)], ["s%d" % i for i in range(nx1)], A1, B1)
    s2 = api.ControlSystem(["v%d" % j for j in range(nu2)], ["r%d" % i for i in range(nx2)], A2, B2)
    e1 = api.ExtendedSystem.from_cotrol_system(s1, "x", N)
    e2 = api.ExtendedSystem.from_cotrol_system(s2, "z", N)
    form = api.Formulation()
    form.incorporate_dynamics("first", e1)
    form.incorporate_dynamics("second", e2)
    for i in range(nx1):
        form.incorporate_goal("track s%d" % i, api.Cost("s%d" % i, float(rng.uniform(0.1, 1)), aim=[float(rng.normal())]))
    for i in range(nx2):
        form.incorporate_goal("track r%d" % i, api.Cost("r%d" % i, float(rng.uniform(0.1, 1)), aim=[float(rng.normal())]))
    form.incorporate_goal("effort", api.Cost("u0", 0.3))
    form.incorporate_constraint("bounds", [api.Constraint("s0", 4.0), api.Constraint("r0", 3.0, arrow=[-1]),
                                           api.Constraint("r1", 2.0)])
    optim = ["u%d" % j for j in range(nu1)] + (["v%d" % j for j in range(nu2)] if optim_second else [])
    form.identify_qp_domain(optim)
    form.make_preview_matrices()
    return form, (A1, B1), (A2, B2)


@pytest.mark.parametrize("lti,optim_second", [((), True), (("first", "second"), True), (("first",), False),
                                              (("first", "second"), False)])
def test_two_plants_in_one_wide_problem(gpu_api, torch_gpu, lti, optim_second):
    """Two systems side by side: columns of one plant's inputs are structural zeros in the rows of
    the other's states (tile masks per 64-column quarter), two generated groups (the general form of
    the kernel: the Toeplitz form is for a single group), and inputs that are GIVEN instead of
    unknown (their horizon tables are read through the given columns of the column tables)."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(31)
    N, nx1, nu1, nx2, nu2 = 48, 3, 3, 2, 2
    form, ab1, ab2 = _two_systems(gpu_api, rng, N, nx1, nu1, nx2, nu2, optim_second)
    assert form.optim_len == N * (nu1 + (nu2 if optim_second else 0))
    B = 11
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    til = engine.Assembler(form, batch=B, lti=list(lti))
    assert til.plan.itab[_H["T_OK"]] == 1
    for name, (A, Bm) in (("first", ab1), ("second", ab2)):
        if name in lti:
            til.bind_lti(name, torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
    out = tuple(torch.full_like(t, float("nan")) for t in til.assemble(given))
    Pt, qt, Gt, ht = (t.clone() for t in til.assemble(given, out=out))
    assert not any(torch.isnan(t).any().item() for t in (Pt, qt, Gt, ht))
    assert "tiled" in til.last_kernel()
    ref = engine.Assembler(form, batch=B)
    ref.set_option(capi.OPT_PATH, 2)
    Ps, qs, Gs, hs = ref.assemble(given)
    assert max(_rel(Pt, Ps), _rel(qt, qs), _rel(Gt, Gs), _rel(ht, hs)) <= 1e-12
    for b in (0, B - 1):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
        assert_close(Pt[b].cpu().numpy(), Qo, RTOL), assert_close(qt[b].cpu().numpy(), qo.ravel(), RTOL)
        assert_close(Gt[b].cpu().numpy(), Ao, RTOL), assert_close(ht[b].cpu().numpy(), ho.ravel(), RTOL)
    # the preview of every definition through the same tables
    x = torch.as_tensor(rng.normal(0, 0.5, [B, form.optim_len]), device="cuda")
    rows = til.preview_rows(given, x).cpu().numpy()
    PM = orc.preview_matrices(form)
    for var, (r0, n) in til.plan.pm_rows.items():
        want = (PM[var][0] @ given[3].cpu().numpy().reshape(-1, 1) + PM[var][1] @ x[3].cpu().numpy().reshape(-1, 1)).ravel()
        assert_close(rows[3, r0:r0 + n], want, 1e-11, var)


@pytest.mark.parametrize("nx,nu,N", [(3, 3, 43), (2, 5, 27), (4, 1, 131)], ids=["129", "135", "131"])
def test_odd_width_on_the_tiled_kernel(gpu_api, torch_gpu, nx, nu, N):
    """An odd number of unknowns (129, 135, 131): every other row of G starts 8 bytes off a 16-byte
    boundary, so the general form of the tiled kernel writes such rows with 8-byte stores and the last
    column alone (round 3 sent these plans to the staged pipeline).  Every instance against the oracle and
    against the staged pipeline; with the horizon tables generated from the system's own (A, B) the
    general form reads them (the Toeplitz forms write 16-byte pieces: even widths only)."""
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(5)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    assert form.optim_len == nu * N and form.optim_len % 2 == 1
    B = 5
    asm = engine.Assembler(form, batch=B)
    assert asm.plan.itab[_H["T_OK"]] == 1 and asm.plan.itab[_H["T_CI_OK"]] == 1
    given = rng.normal(0, 0.3, [B, form.given_len])
    out = [torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")
           for shape in ((B, asm.no, asm.no), (B, asm.no), (B, asm.nc, asm.no), (B, asm.nc))]
    P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given, out=tuple(out)))
    assert asm.last_kernel() == "tiled_assemble_kernel", asm.last_kernel()
    for b in range(B):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
        assert_close(P[b], Qo, RTOL_TIGHT), assert_close(q[b], qo.ravel(), RTOL_TIGHT)
        assert_close(G[b], Ao, RTOL_TIGHT), assert_close(h[b], ho.ravel(), RTOL_TIGHT)
    asm.set_option(capi.OPT_PATH, 2)
    staged = [t.cpu().numpy() for t in asm.assemble(given)]
    assert "staged" in asm.last_kernel()
    for mine, theirs in zip((P, q, G, h), staged):
        assert_close(mine, theirs, RTOL_TIGHT)
    # a system of its own per instance: tables from the pre-pass, read by the general form; against the staged
    # pipeline on the horizon matrices mpcasm_fill_su writes for the same systems
    As, Bs = zip(*(problems.random_lti_matrices(rng, nx, nu) for _ in range(B)))
    At, Bt = torch.as_tensor(np.stack(As), device="cuda"), torch.as_tensor(np.stack(Bs), device="cuda")
    lti = engine.Assembler(form, batch=B, lti=["plant"])
    lti.bind_lti("plant", At, Bt)
    res = [t.cpu().numpy() for t in lti.assemble(given)]
    # (a narrow system with its tables on chip may fit the persistent kernel instead: nx = 2)
    assert lti.last_kernel() == "tiled_assemble_kernel" or "persistent" in lti.last_kernel(), lti.last_kernel()
    S, U = engine.fill_su(At, Bt, N)
    for j in range(nu):
        asm.bind_source(("plant", j), U[:, j])
    asm.bind_source(("plant", nu), S)
    ref = [t.cpu().numpy() for t in asm.assemble(given)]
    assert "staged" in asm.last_kernel()
    for mine, theirs in zip(res, ref):
        assert_close(mine, theirs, RTOL_TIGHT)


def test_a_non_causal_matrix_is_refused_where_the_masks_assume_causality(gpu_api, torch_gpu):
    """the tiled kernel skips the tiles of U_j above the diagonal.  A tensor with entries
    there is refused by bind_source (ValueError) and by rebind_sources (False: recompile), instead of
    silently giving a wrong P."""
    torch = torch_gpu
    from mpcasm import engine

    rng = np.random.default_rng(8)
    form = problems.random_lti(gpu_api, rng, nx=4, nu=4, N=32)
    asm = engine.Assembler(form, batch=3)
    assert asm.plan.causal_assumed
    dyn = form.dynamics["plant"]
    U1 = np.array(dyn.matrices[1])
    bad = np.broadcast_to(U1, (3,) + U1.shape).copy()
    bad[1, 3, 20, 0] = 0.25
    with pytest.raises(ValueError):
        asm.bind_source(("plant", 1), torch.as_tensor(bad, device="cuda"))
    asm.bind_source(("plant", 1), torch.as_tensor(np.broadcast_to(U1, (3,) + U1.shape).copy(), device="cuda"))
    frozen = {("plant", k): np.array(M) for k, M in enumerate(dyn.matrices)}
    assert asm.rebind_sources(form, frozen)
    frozen[("plant", 1)][3, 20, 0] = 0.25
    assert not asm.rebind_sources(form, frozen)
