"""Randomised formulations: the plan compiler + kernels against the oracle over many
structures the fixed fixtures do not reach (random numbers of axes, inputs, costs with
and without L / schedule / cross terms, constraints with per-row fields, boxes).

CPU: plan tables through the numpy emulator.  GPU: the three kernel paths.
The generator is seeded; every case is also a regression test."""
import numpy as np
import pytest

import plan_emulator
from helpers import RTOL_TIGHT, assert_close
from mpcasm import problems
from mpcasm.plan import compile_plan
from oracle import qp_oracle as orc


def random_formulation(api, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 4))
    m = int(rng.integers(1, 3))
    N = int(rng.integers(2, 9))
    axes_pool = [None, ["_x"], ["_x", "_y"], ["_x", "_y", "_z"]]
    axes = axes_pool[int(rng.integers(0, 4))]
    ax = axes or [""]
    A, B = problems.random_lti_matrices(rng, n, m)
    inputs = ["u%d" % j for j in range(m)]
    states = ["s%d" % i for i in range(n)]
    ext = api.ExtendedSystem.from_cotrol_system(
        api.ControlSystem(inputs, states, A, B, axes), "x", N)
    extra = api.DomainVariable("w", N, axes)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    form.incorporate_dynamics("extra", extra)
    # derived definitions: scalar, 1-D and 2-D coefficients
    rows2 = int(rng.integers(1, N + 1))
    last = states[-1]
    for a in ax:
        form.incorporate_definition("mix" + a, api.LineCombo(
            {"s0" + a: float(rng.normal()), "w" + a: rng.standard_normal((N, N))}))
        form.incorporate_definition("sel" + a, api.LineCombo(
            {"mix" + a: rng.standard_normal((rows2, N)),
             last + a: rng.standard_normal((rows2, N))}))
        form.incorporate_definition("avg" + a, api.LineCombo({"s0" + a: rng.standard_normal(N)}))

    def sched(rows):
        if rows < 2 or rng.random() < 0.5:
            return None, rows
        lo = int(rng.integers(0, rows - 1))
        hi = int(rng.integers(lo + 1, rows + 1))
        return range(lo, hi), hi - lo

    var_rows = {"s0": N, "mix": N, "sel": rows2, "avg": 1, "w": N, last: N, "u0": N}
    names = list(var_rows)
    for k in range(int(rng.integers(1, 5))):
        var = names[int(rng.integers(0, len(names)))]
        count = int(rng.integers(1, len(ax) + 1))
        use_axes = [ax[i] for i in sorted(rng.choice(len(ax), count, replace=False))]
        s, t = sched(var_rows[var])
        L = None
        if rng.random() < 0.4:
            L = [rng.standard_normal((int(rng.integers(1, 4)), t))]
        kwargs = dict(aim=list(rng.normal(size=len(use_axes))), axes=use_axes if axes else None,
                      L=L, schedule=s)
        if rng.random() < 0.3 and L is None:
            cross = [v for v in names if var_rows[v] == var_rows[var] and v != var]
            if cross:
                kwargs.update(cross=cross[0], cross_aim=list(rng.normal(size=len(use_axes))))
        form.incorporate_goal("cost%d" % k, api.Cost(var, float(rng.uniform(0, 2)), **kwargs))
    for k in range(int(rng.integers(1, 4))):
        var = names[int(rng.integers(0, len(names)))]
        s, t = sched(var_rows[var])
        na = len(ax)
        style = int(rng.integers(0, 3))
        if style == 0:      # one row of geometry, broadcast over the variable's rows
            c = api.Constraint(var, float(rng.uniform(0.5, 2)), axes=axes,
                               arrow=list(rng.normal(size=na)), center=list(rng.normal(size=na)),
                               schedule=s)
        elif style == 1:    # per-row geometry
            c = api.Constraint(var, rng.uniform(0.5, 2, t), axes=axes,
                               arrow=rng.normal(size=(t, na)), center=rng.normal(size=(t, na)),
                               schedule=s)
        else:               # rows from L
            mrows = int(rng.integers(1, 4))
            c = api.Constraint(var, float(rng.uniform(0.5, 2)), axes=axes,
                               arrow=list(rng.normal(size=na)),
                               L=[rng.standard_normal((mrows, t)) for _ in range(na)], schedule=s)
        form.incorporate_constraint("limit%d" % k, c)
    if axes and len(axes) >= 2 and rng.random() < 0.7:
        form.incorporate_box("box", api.Box.task_space(
            "s0", api.tools.make_simetric_vertices(rng.uniform(0.5, 1.5, 2)), axes[:2]))
    optim = [v for v in form.domain if v.startswith("u")]
    if rng.random() < 0.3:
        optim += [v for v in form.domain if v.startswith("w")]
    form.identify_qp_domain(optim)
    form.make_preview_matrices()
    return form, rng


SEEDS = list(range(24))


@pytest.mark.parametrize("seed", SEEDS)
def test_plan_tables_random(cpu_api, seed):
    form, rng = random_formulation(cpu_api, seed)
    given = rng.standard_normal([form.given_len, 1])
    plan = compile_plan(form)
    out = plan_emulator.run(plan, given)
    A, h, Q, q = orc.assemble(form, given)
    assert_close(out["P"], Q, 1e-12)
    assert_close(out["q"], q.ravel(), 1e-12)
    assert_close(out["G"], A, 1e-12)
    assert_close(out["h"], h.ravel(), 1e-12)
    if plan.itab[plan_emulator.H["FUSED_OK"]]:
        assert_close(plan_emulator.run_fused_workspace(plan, given), out["V"], 1e-12)
    if plan.itab[plan_emulator.H["RS_OK"]]:
        res = plan_emulator.run_resident(plan, given)
        assert_close(res["P"], Q, 1e-12)
        assert_close(res["q"], q.ravel(), 1e-12)
        assert_close(res["G"], A, 1e-12)
        assert_close(res["h"], h.ravel(), 1e-12)
        # the CSC form of the same results, where the row records allow it (at most two axes):
        # crossed costs make P unsymmetric, so all of it is stored
        try:
            sparse = compile_plan(form, csc="full")
        except ValueError:
            return
        got = plan_emulator.run_resident(sparse, given)
        c = sparse.csc
        assert ((Q != 0) <= sparse.P_pattern).all() and ((A != 0) <= sparse.G_pattern).all()
        assert_close(got["P_data"], Q.ravel()[c["p_flat"]], 1e-12)
        assert_close(got["G_data"], A.ravel()[c["g_flat"]], 1e-12)
        assert (np.diff(c["P"][0]) >= 0).all() and c["P"][0][-1] == c["pnnz"] == c["p_flat"].size
        assert (np.diff(c["G"][0]) >= 0).all() and c["G"][0][-1] == c["gnnz"] == c["g_flat"].size


@pytest.mark.gpu
@pytest.mark.parametrize("path", [0, 1, 2])
def test_kernels_random(gpu_api, path):
    from mpcasm import capi
    from mpcasm.engine import Assembler

    lib = capi.load()
    assert lib.mpcasm_set_option(capi.OPT_PATH, path) == 0
    try:
        for seed in SEEDS:
            form, rng = random_formulation(gpu_api, seed)
            batch = 5
            given = rng.standard_normal([batch, form.given_len])
            asm = Assembler(form, batch=batch)
            P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
            PM = orc.preview_matrices(form)
            for b in (0, batch - 1):
                A, hh, Q, qq = orc.assemble(form, given[b].reshape(-1, 1), PM)
                assert_close(P[b], Q, RTOL_TIGHT, "P seed %d" % seed)
                assert_close(q[b], qq.ravel(), RTOL_TIGHT, "q seed %d" % seed)
                assert_close(G[b], A, RTOL_TIGHT, "G seed %d" % seed)
                assert_close(h[b], hh.ravel(), RTOL_TIGHT, "h seed %d" % seed)
            pm = asm.preview_matrices()[0].cpu().numpy()
            for var, (r0, rows) in asm.plan.pm_rows.items():
                assert_close(pm[r0:r0 + rows, :asm.ng], PM[var][0], RTOL_TIGHT, var)
                assert_close(pm[r0:r0 + rows, asm.ng:], PM[var][1], RTOL_TIGHT, var)
    finally:
        lib.mpcasm_set_option(capi.OPT_PATH, 0)


@pytest.mark.gpu
def test_persistent_kernel_instance_loop(gpu_api):
    """Several instances per workgroup (double-buffered input images, clearing of shared
    workspace elements, P / q reuse): every instance of the persistent kernel against the
    staged pipeline, with per-instance parameters."""
    import torch

    from mpcasm import capi
    from mpcasm.engine import Assembler

    lib = capi.load()
    batch = 1300
    try:
        for seed in SEEDS[:12]:
            form, rng = random_formulation(gpu_api, seed)
            given = rng.standard_normal([batch, form.given_len])
            asm = Assembler(form, batch=batch)
            asm.params[:] = asm.params * torch.as_tensor(
                rng.uniform(0.5, 1.5, tuple(asm.params.shape)), device=asm.params.device)
            out = {}
            for path in (2, 0):
                assert lib.mpcasm_set_option(capi.OPT_PATH, path) == 0
                out[path] = [t.cpu().numpy().copy() for t in asm.assemble(given)]
            for a, b, name in zip(out[0], out[2], "PqGh"):
                assert_close(a, b, 1e-12, "%s seed %d" % (name, seed))
    finally:
        lib.mpcasm_set_option(capi.OPT_PATH, 0)


@pytest.mark.gpu
def test_csc_form_random(gpu_api):
    """Random formulations whose problems fit on chip: the CSC data the persistent kernel writes
    (plans compiled with csc='full') are, entry for entry, the dense results of the same inputs."""
    from mpcasm.engine import Assembler

    batch, done = 700, 0
    for seed in SEEDS:
        form, rng = random_formulation(gpu_api, seed)
        given = rng.standard_normal([batch, form.given_len])
        try:
            sparse = Assembler(form, batch=batch, csc="full")
        except (ValueError, RuntimeError):
            continue                         # no persistent kernel for this problem
        dense = Assembler(form, batch=batch)
        Pd, q, Gd, h = (t.cpu().numpy() for t in sparse.assemble(given))
        P, q2, G, h2 = (t.cpu().numpy() for t in dense.assemble(given))
        c = sparse.csc
        assert_close(Pd, P.reshape(batch, -1)[:, c["p_flat"]], 1e-13, "P seed %d" % seed)
        assert_close(Gd, G.reshape(batch, -1)[:, c["g_flat"]], 1e-13, "G seed %d" % seed)
        assert_close(q, q2, 1e-13, "q seed %d" % seed)
        assert_close(h, h2, 1e-13, "h seed %d" % seed)
        done += 1
    assert done >= 6, done
