"""GPU: the assembly kernels (K2 + K3 + K4 through mpcasm_assemble /
mpcasm_preview_matrices) against the golden vectors of the real reference and the
oracle.  Index maps bit-exact; P, q, G, h and the preview matrices within 1e-10
relative (north star), observed ~1e-15.
"""
import json

import numpy as np
import pytest

from helpers import RTOL, RTOL_TIGHT, assert_close, golden, ranges_from_json
from mpcasm import problems
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


# path name -> (MPCASM_OPT_PATH, MPCASM_OPT_JIT): the persistent kernel ahead of time and
# compiled per plan by hiprtc (forced on for every batch size here), the per-instance fused
# kernel, the staged pipeline
PATHS = {"resident": (0, 2, 1), "resident-jit": (0, 1, 1), "resident-lds": (0, 2, 2),
         "resident-jit-lds": (0, 1, 2), "fused": (1, 2, 0), "staged": (2, 2, 0)}  # (..., MPCASM_OPT_P_DIRECT)
RESIDENT = ("resident", "resident-jit", "resident-lds", "resident-jit-lds")


@pytest.fixture(autouse=True, params=list(PATHS))
def kernel_path(request):
    """Every test runs on each assembly path: the persistent fused kernel (ahead of time,
    specialised for the plan at run time, and with P collected in LDS instead of stored block by
    block), the per-instance fused kernel (all taken only when one instance fits on chip) and
    the staged K2 -> K3 -> K4 pipeline."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mpcasm import capi

    lib = capi.load()
    assert lib.mpcasm_set_option(capi.OPT_PATH, PATHS[request.param][0]) == 0
    assert lib.mpcasm_set_option(capi.OPT_JIT, PATHS[request.param][1]) == 0
    assert lib.mpcasm_set_option(capi.OPT_P_DIRECT, PATHS[request.param][2]) == 0
    yield request.param
    lib.mpcasm_set_option(capi.OPT_PATH, 0)
    lib.mpcasm_set_option(capi.OPT_JIT, 0)
    lib.mpcasm_set_option(capi.OPT_P_DIRECT, 0)


def check_drop_in(form, g, prefix, parts=True):
    """The reference's call sequence on this repository's Formulation (B=1 path)."""
    assert {k: form.optim_ID[k] for k in form.optim_variables} == \
        ranges_from_json(g[prefix + "optim_ID"])
    assert {k: form.given_ID[k] for k in form.given_variables} == \
        ranges_from_json(g[prefix + "given_ID"])
    given = g[prefix + "given"]
    names = json.loads(str(g[prefix + "definitions"]))
    assert list(form.PM.keys()) == names and len(form.PM) == len(form.definitions)
    for var in names:
        if prefix + "PM/" + var + "/Mg" in g:
            Mg, Mo = form.PM[var]
            assert_close(Mg, g[prefix + "PM/" + var + "/Mg"], RTOL_TIGHT, var + " Mg")
            assert_close(Mo, g[prefix + "PM/" + var + "/Mo"], RTOL_TIGHT, var + " Mo")
    if parts:
        for k, limit in enumerate(orc.all_limits(form)):
            if prefix + "limit%d/A" % k in g:
                A, h = form.generate_qp_constraint(limit, given)
                assert_close(A, g[prefix + "limit%d/A" % k], RTOL_TIGHT, "limit A")
                assert_close(h, g[prefix + "limit%d/h" % k], RTOL_TIGHT, "limit h")
        for name, cost in form.goals.items():
            if prefix + "cost/" + name + "/Q" in g:
                Q, q = form.generate_qp_cost(cost, given)
                assert_close(Q, g[prefix + "cost/" + name + "/Q"], RTOL_TIGHT, "cost Q " + name)
                assert_close(q, g[prefix + "cost/" + name + "/q"], RTOL_TIGHT, "cost q " + name)
    A, h, Q, q = form.generate_all_qp_matrices(given)
    assert h.ndim == 2 and h.shape[1] == 1 and q.shape == (form.optim_len, 1)
    for mine, nm in ((A, "A"), (h, "h"), (Q, "Q"), (q, "q")):
        assert mine.dtype == np.float64 and mine.flags["C_CONTIGUOUS"]
        assert_close(mine, g[prefix + nm], RTOL_TIGHT, prefix + nm)
    A2, h2 = form.generate_all_qp_constraints(given)
    Q2, q2 = form.generate_all_qp_costs(given)
    assert np.array_equal(A2, A) and np.array_equal(h2, h)
    assert np.array_equal(Q2, Q) and np.array_equal(q2, q)
    return A, h, Q, q


def test_body_case_drop_in(gpu_api):
    """The problem of the reference's test_body.py: shapes it asserts (:140-162)
    plus the values captured from the reference."""
    g = golden("g2_body")
    form = problems.body_case(gpu_api)
    A, h, Q, q = check_drop_in(form, g, "arange/")
    assert A.shape == (54, form.optim_len) and h.shape == (54, 1)
    limit = form.constraints["kinematics"][0]
    A1, h1 = form.generate_qp_constraint(limit, g["arange/given"])
    assert A1.shape == (9, form.optim_len) and (h1 == limit.bound()).all()   # test_body.py:142
    check_drop_in(form, g, "random/")
    assert np.abs(Q - Q.T).max() > 1e-6          # crossed cost: non-symmetric Hessian


@pytest.mark.parametrize("step_samples", [8, 12])
def test_biped_ticks_drop_in(gpu_api, step_samples):
    """C1: single instance, the walking loop's update -> arrange_given ->
    generate_all_qp_matrices sequence (biped_mpc_loop.py:51-56), both QP widths."""
    conf = problems.BipedConfig(step_samples=step_samples)
    g = golden("g3_biped_N%d" % conf.horizon_lenght)
    form = problems.biped(gpu_api, conf)
    clock = problems.StepClock(conf.step_samples, form.domain["Ds_x"])
    keep = [int(t) for t in g["ticks"]]
    for tick in range(18):
        form.update(step_times=clock.step_times, step_count=clock.step_count)
        if tick in keep:
            A, h, Q, q = check_drop_in(form, g, "tick%02d/" % tick, parts=(tick == keep[0]))
            row = g["shapes"][tick]
            assert (Q.shape[0], A.shape[0]) == (row[2], row[3])
        clock.tick()


def test_numbers_changed_between_calls_are_picked_up(gpu_api):
    """Cost / Constraint objects are mutated in place between calls in the
    reference's usage; the cached plan must re-read them."""
    form = problems.body_case(gpu_api)
    given = np.random.default_rng(0).standard_normal([form.given_len, 1])
    form.generate_all_qp_matrices(given)
    form.goals["velocity"].update(aim=[3.0, -1.0], weight=7.0)
    form.constraints["kinematics"][0].update(extreme=2.5)
    form.goals["terminal"].update(schedule=range(6, 9))          # structure change
    A, h, Q, q = form.generate_all_qp_matrices(given)
    Ao, ho, Qo, qo = orc.assemble(form, given)
    for mine, ref in ((A, Ao), (h, ho), (Q, Qo), (q, qo)):
        assert_close(mine, ref, RTOL_TIGHT)


def test_preview_and_goal_distance(gpu_api):
    form = problems.body_case(gpu_api)
    rng = np.random.default_rng(1)
    given = rng.standard_normal([form.given_len, 1])
    optim = rng.standard_normal([form.optim_len, 1])
    PM = orc.preview_matrices(form)
    for var in ("CoM_x", "DCM_y", "s_x"):
        assert_close(form.preview(given, optim, var), orc.preview(PM, given, optim, var),
                     RTOL_TIGHT, var)
    both = form.preview(given, optim, "CoM", ["_x", "_y"])
    assert both.shape == (9, 2)
    assert_close(both, orc.preview(PM, given, optim, "CoM", ["_x", "_y"]), RTOL_TIGHT)
    v = orc.preview(PM, given, optim, "DCM_x") - form.goals["stability"].aim[:, 0]
    assert abs(form.goal_distance(given, optim, "stability") - float(v.T @ v)) < 1e-9


def batched_case(form, batch, seed, vary):
    """Assemble `batch` instances with per-instance given and parameters; check a
    sample of instances one by one against the oracle run on a mutated copy of
    the description."""
    import torch
    from mpcasm.engine import Assembler

    rng = np.random.default_rng(seed)
    asm = Assembler(form, batch=batch)
    given = rng.normal(0, 0.1, [batch, form.given_len])
    overrides = vary(asm, rng, batch)
    P, q, G, h = asm.assemble(torch.as_tensor(given, device="cuda"))
    torch.cuda.synchronize()
    P, q, G, h = (t.cpu().numpy() for t in (P, q, G, h))
    assert P.shape == (batch, form.optim_len, form.optim_len) and h.shape[0] == batch
    return asm, given, overrides, (P, q, G, h)


def test_biped_batch_c2(gpu_api):
    """C2: B=4096 instances of the biped (36-wide phase), per-instance given
    vector, velocity aim and stepping-area centres."""
    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)      # phi = 1: 36-wide
    assert form.optim_len == 36
    limits = orc.all_limits(form)
    batch = 4096

    def vary(asm, rng, B):
        aims = rng.uniform(0, 0.6, [B, 1, 1])
        asm.set_param("cost", "track vel_x", "aim", aims)
        centers = {}
        for k in range(4):                                      # facets of the stepping area
            c = np.asarray(limits[k].center, dtype=float)[None] + rng.normal(0, 0.01, [B, 2, 2])
            asm.set_param("limit", k, "center", c)
            centers[k] = c
        return aims, centers

    asm, given, (aims, centers), (P, q, G, h) = batched_case(form, batch, 20260, vary)
    assert G.shape == (batch, 76, 36)
    # 64 instances against the oracle: the ends, the instances around the boundaries of the workgroups'
    # runs of four, random ones (four were thin where the oracle costs a millisecond each)
    picks = sorted({0, 1, 3, 4, 2047, 2048, 1777, batch - 2, batch - 1}
                   | set(int(x) for x in np.random.default_rng(4).integers(0, batch, 55)))
    for b in picks:
        form.goals["track vel_x"].update(aim=aims[b, 0])
        for k in range(4):
            limits[k].update(center=centers[k][b])
        A, hh, Q, qq = orc.assemble(form, given[b].reshape(-1, 1))
        assert_close(P[b], Q, RTOL_TIGHT, "P")
        assert_close(q[b], qq.ravel(), RTOL_TIGHT, "q")
        assert_close(G[b], A, RTOL_TIGHT, "G")
        assert_close(h[b], hh.ravel(), RTOL_TIGHT, "h")
    # size-independent property over the whole batch: P does not depend on given /
    # aims here, so it is the same matrix for every instance, and symmetric
    assert np.abs(P - P[0]).max() == 0.0
    assert np.abs(P[0] - P[0].T).max() <= 1e-12 * np.abs(P[0]).max()
    # ... G does not depend on given either, and q, h are affine in it, for EVERY instance (its own
    # aims and centres): f(g) + f(g') = f(g + g') + f(0)
    import torch

    rng = np.random.default_rng(77)
    g1 = rng.normal(0, 0.1, given.shape)
    _, q1, G1, h1 = (t.cpu().numpy().copy() for t in asm.assemble(g1))
    _, q0, _, h0 = (t.cpu().numpy().copy() for t in asm.assemble(np.zeros_like(g1)))
    _, qs, Gs, hs = (t.cpu().numpy().copy() for t in asm.assemble(given + g1))
    assert np.array_equal(G1, G) and np.array_equal(Gs, G)
    assert np.abs((q + q1) - (qs + q0)).max() <= 1e-12 * np.abs(qs).max()
    assert np.abs((h + h1) - (hs + h0)).max() <= 1e-12 * np.abs(hs).max()


def test_biped_batch_per_instance_dynamics(gpu_api):
    """Per-instance horizon matrices: K1's output bound as a source of K2."""
    import torch
    from mpcasm import engine

    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([7, 15]), step_count=0)      # phi = 0: 34-wide
    assert form.optim_len == 34
    batch = 33
    rng = np.random.default_rng(5)
    get_A, get_B, _ = gpu_api.tools.get_system_matrices("J->CCC")
    taus = rng.uniform(0.08, 0.12, batch)
    A = np.stack([get_A(tau=t) for t in taus])
    B = np.stack([get_B(tau=t) for t in taus])
    S, U = engine.fill_su(A, B, conf.horizon_lenght)
    asm = engine.Assembler(form, batch=batch)
    assert ("LIP", 0) in asm.source_keys() and ("LIP", 1) in asm.source_keys()
    asm.bind_source(("LIP", 0), U[:, 0])
    asm.bind_source(("LIP", 1), S)
    given = rng.normal(0, 0.1, [batch, form.given_len])
    P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
    lip = form.dynamics["LIP"]
    for b in (0, 7, batch - 1):
        Sb, Ub = orc.extend_matrices(conf.horizon_lenght, A[b], B[b])
        lip.matrices = Ub + [Sb]
        lip.update_definitions()
        Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
        assert_close(P[b], Qo, RTOL_TIGHT)
        assert_close(q[b], qo.ravel(), RTOL_TIGHT)
        assert_close(G[b], Ao, RTOL_TIGHT)
        assert_close(h[b], ho.ravel(), RTOL_TIGHT)


def test_biped_batch_horizon_matrices_generated_on_chip(gpu_api, kernel_path):
    """K1 fused into the assembly: the plan is compiled with ``lti=["LIP"]``, the kernel gets
    per-instance ``(A, B)`` and builds what it needs of ``S, U`` in LDS.  Same numbers as
    fill_su + assemble; both QP widths; shared and per-instance systems."""
    from mpcasm import engine

    if kernel_path not in RESIDENT:
        pytest.skip("generated sources exist in the persistent kernel only")
    get_A, get_B, _ = gpu_api.tools.get_system_matrices("J->CCC")
    for times, width in (([7, 15], 34), ([6, 14], 36)):
        conf = problems.BipedConfig(step_samples=8)
        form = problems.biped(gpu_api, conf)
        form.update(step_times=np.array(times), step_count=0)
        assert form.optim_len == width
        batch = 300 if width == 36 else 5
        rng = np.random.default_rng(width)
        given = rng.normal(0, 0.1, [batch, form.given_len])
        asm = engine.Assembler(form, batch=batch, lti=["LIP"])
        assert asm.plan.resident["ok"] and asm.plan.lti[0]["n"] == 3
        # shared system: (A, B) recovered from the formulation's own S, U
        ref = engine.Assembler(form, batch=batch)
        for mine, theirs in zip(asm.assemble(given), ref.assemble(given)):
            assert_close(mine.cpu().numpy(), theirs.cpu().numpy(), RTOL_TIGHT)
        with pytest.raises(ValueError):
            asm.bind_source(("LIP", 0), np.zeros((16, 16, 3)))
        # one system per instance
        taus = rng.uniform(0.08, 0.12, batch)
        A = np.stack([get_A(tau=t) for t in taus])
        B = np.stack([get_B(tau=t) for t in taus])
        asm.bind_lti("LIP", A, B)
        P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
        lip = form.dynamics["LIP"]
        keep = list(lip.matrices)
        for b in (0, 3, batch - 1):
            Sb, Ub = orc.extend_matrices(conf.horizon_lenght, A[b], B[b])
            lip.matrices = Ub + [Sb]
            lip.update_definitions()
            Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
            assert_close(P[b], Qo, RTOL_TIGHT)
            assert_close(q[b], qo.ravel(), RTOL_TIGHT)
            assert_close(G[b], Ao, RTOL_TIGHT)
            assert_close(h[b], ho.ravel(), RTOL_TIGHT)
        lip.matrices = keep
        lip.update_definitions()
        # one half of the outputs only: the tables are still built
        Gc, hc = asm.assemble(given, want_cost=False)[2:]
        assert np.array_equal(Gc.cpu().numpy(), G) and np.array_equal(hc.cpu().numpy(), h)
        Pc, qc = asm.assemble(given, want_constraints=False)[:2]
        assert np.array_equal(Pc.cpu().numpy(), P) and np.array_equal(qc.cpu().numpy(), q)


def test_result_buffers_must_be_16_byte_aligned(gpu_api):
    """The kernels store results 16 bytes at a time: a result buffer that starts 8 bytes off
    is refused (MPCASM_ERR_ARG), not written through a misaligned pointer."""
    import torch

    from mpcasm import capi, engine

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    batch = 4
    asm = engine.Assembler(form, batch=batch)
    given = np.random.default_rng(3).normal(0, 0.1, [batch, form.given_len])
    no, nc = asm.no, asm.nc
    f = dict(dtype=torch.float64, device="cuda")
    sizes = (batch * no * no, batch * no, batch * nc * no, batch * nc)
    shapes = ((batch, no, no), (batch, no), (batch, nc, no), (batch, nc))
    ref = [t.cpu().numpy() for t in asm.assemble(given)]
    for bad in range(4):
        bufs = [torch.empty(n + 1, **f) for n in sizes]
        out = [b[1:].view(shape) if i == bad else b[:-1].view(shape)
               for i, (b, shape) in enumerate(zip(bufs, shapes))]
        with pytest.raises(capi.MpcasmError) as err:
            asm.assemble(given, out=tuple(out))
        assert err.value.status == -1          # MPCASM_ERR_ARG
    bufs = [torch.empty(n + 2, **f) for n in sizes]             # 16 bytes off: fine
    out = tuple(b[2:].view(shape) for b, shape in zip(bufs, shapes))
    for mine, theirs in zip(asm.assemble(given, out=out), ref):
        assert np.array_equal(mine.cpu().numpy(), theirs)


def test_workgroups_per_cu_option_changes_nothing_but_the_grid(gpu_api):
    """MPCASM_OPT_RESIDENT_PER_CU (tuning aid): one persistent workgroup per CU walks more
    instances each; same bits out (every instance is assembled by one workgroup alone)."""
    from mpcasm import capi, engine

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    batch = 700
    asm = engine.Assembler(form, batch=batch)
    given = np.random.default_rng(11).normal(0, 0.1, [batch, form.given_len])
    ref = [t.cpu().numpy().copy() for t in asm.assemble(given)]
    lib = capi.load()
    try:
        for per_cu in (1, 2):
            assert lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, per_cu) == 0
            for mine, theirs in zip(asm.assemble(given), ref):
                assert np.array_equal(mine.cpu().numpy(), theirs)
        assert lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, -1) == -1
    finally:
        lib.mpcasm_set_option(capi.OPT_RESIDENT_PER_CU, 0)


def test_a_share_of_the_workgroup_slots_changes_nothing_but_the_grid(gpu_api):
    """MPCASM_OPT_RESIDENT_GRID, per plan: a launch that may take 48 (or 3) workgroups walks more
    instances with each; same bits out, on the compiled and the ahead-of-time kernel."""
    from mpcasm import capi, engine

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    batch = 700
    asm = engine.Assembler(form, batch=batch)
    given = np.random.default_rng(12).normal(0, 0.1, [batch, form.given_len])
    ref = [t.cpu().numpy().copy() for t in asm.assemble(given)]
    for jit in (1, 2):
        asm.set_option(capi.OPT_JIT, jit)
        for grid in (48, 3, 0):
            asm.set_option(capi.OPT_RESIDENT_GRID, grid)
            for mine, theirs in zip(asm.assemble(given), ref):
                assert np.array_equal(mine.cpu().numpy(), theirs)
    with pytest.raises(capi.MpcasmError):
        asm.set_option(capi.OPT_RESIDENT_GRID, -2)


def test_biped_long_horizon_persistent_kernel(gpu_api):
    """N = 24 (no = 52, nc = 108): more 16-byte pieces of G than the per-thread descriptor
    table of the persistent kernel holds, so G goes by the packed words of the row records
    (plan_tables.h RR_PACKED); 13 block columns of P and q.  Horizon matrices built on chip,
    against the staged pipeline fed by fill_su."""
    from mpcasm import capi, engine

    conf = problems.BipedConfig(step_samples=12)
    form = problems.biped(gpu_api, conf)
    clock = problems.StepClock(conf.step_samples, form.domain["Ds_x"])
    form.update(step_times=clock.step_times, step_count=clock.step_count)
    batch = 70
    rng = np.random.default_rng(24)
    given = rng.normal(0, 0.1, [batch, form.given_len])
    asm = engine.Assembler(form, batch=batch, lti=["LIP"])
    assert asm.plan.resident["ok"] and asm.nc * (asm.no // 2) > 6 * 256
    mine = [t.cpu().numpy() for t in asm.assemble(given)]
    ref = engine.Assembler(form, batch=batch)
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_PATH, 2)
    try:
        theirs = [t.cpu().numpy() for t in ref.assemble(given)]
    finally:
        lib.mpcasm_set_option(capi.OPT_PATH, 0)
    for m, t in zip(mine, theirs):
        assert_close(m, t, RTOL_TIGHT)
    Ao, ho, Qo, qo = orc.assemble(form, given[5].reshape(-1, 1))
    assert_close(mine[0][5], Qo, RTOL_TIGHT)
    assert_close(mine[2][5], Ao, RTOL_TIGHT)


@pytest.mark.parametrize("nx,nu,N", [(2, 2, 6), (4, 1, 9), (5, 2, 5), (6, 3, 7)])
def test_generated_horizon_matrices_other_systems(gpu_api, kernel_path, nx, nu, N):
    """K1 fused for systems other than the 3-state pendulum: several inputs (U_0..U_{m-1}),
    n = 4 (a full quad), and n > 4 (the doubling passes through LDS); horizon lengths that
    are no powers of two.  Per-instance (A, B) against fill_su + assemble and the oracle."""
    from mpcasm import engine

    if kernel_path not in RESIDENT:
        pytest.skip("generated sources exist in the persistent kernel only")
    rng = np.random.default_rng(100 * nx + N)
    form = problems.random_lti(gpu_api, rng, nx=nx, nu=nu, N=N)
    batch = 21
    A = np.stack([problems.random_lti_matrices(rng, nx, nu)[0] for _ in range(batch)])
    B = np.stack([problems.random_lti_matrices(rng, nx, nu)[1] for _ in range(batch)])
    given = rng.normal(0, 1.0, [batch, form.given_len])
    asm = engine.Assembler(form, batch=batch, lti=["plant"])
    assert asm.plan.resident["ok"] and asm.plan.lti[0]["n"] == nx and asm.plan.lti[0]["m"] == nu
    asm.bind_lti("plant", A, B)
    mine = [t.cpu().numpy() for t in asm.assemble(given)]
    S, U = engine.fill_su(A, B, N)
    ref = engine.Assembler(form, batch=batch)
    for j in range(nu):
        ref.bind_source(("plant", j), U[:, j])
    ref.bind_source(("plant", nu), S)
    for x, y in zip(mine, ref.assemble(given)):
        assert_close(x, y.cpu().numpy(), RTOL_TIGHT)
    plant = form.dynamics["plant"]
    Sb, Ub = orc.extend_matrices(N, A[3], B[3])
    plant.matrices = Ub + [Sb]
    plant.update_definitions()
    Ao, ho, Qo, qo = orc.assemble(form, given[3].reshape(-1, 1))
    for x, y in zip(mine, (Qo, qo.ravel(), Ao, ho.ravel())):
        assert_close(x[3], y, RTOL_TIGHT)


def test_batched_box_transforms(gpu_api, kernel_path):
    """f4 (restrictions.py:380-486): recenter, translate, rotate, scale and safety margin of a
    box for every instance at once on the device, against the same calls made instance by
    instance on this repository's host Box and assembled by the oracle."""
    import copy

    from mpcasm import engine
    from mpcasm.boxes import BoxBatch

    if kernel_path not in RESIDENT:
        pytest.skip("the transforms act on the parameters, one assembly path is enough")
    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    batch = 9
    rng = np.random.default_rng(21)
    given = rng.normal(0, 0.1, [batch, form.given_len])
    asm = engine.Assembler(form, batch=batch)
    support = BoxBatch(asm, form, "support_polygon")
    stepping = BoxBatch(asm, form, "stepping area")
    with pytest.raises(KeyError):
        BoxBatch(asm, form, "no such box")

    centers = rng.normal(0, 0.05, [batch, 2])
    shifts = rng.normal(0, 0.02, [batch, 2])
    angles = rng.uniform(-0.5, 0.5, batch)
    rot = np.stack([[[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]] for t in angles])
    scales = rng.uniform(0.6, 1.4, batch)
    margins = rng.uniform(0.0, 0.01, batch)
    stepping.recenter_in_TS(centers)
    stepping.translate_in_TS(shifts)
    support.scale_box(scales)
    support.scale_box(scales * 0.9)          # relative to the factor set before
    # (the margin before the rotation: the host's rotate_in_TS spreads a single arrow over the
    # rows of the facet, after which np.linalg.norm(arrow) is another number)
    support.set_safety_margin(margins)
    support.rotate_in_TS(rot)
    stepping.set_safety_margin(0.002)        # one margin for all
    G, h = (t.cpu().numpy() for t in asm.assemble(given)[2:])

    for b in (0, 4, batch - 1):
        ref = copy.deepcopy(form)
        rs, rp = ref.constraint_boxes["stepping area"], ref.constraint_boxes["support_polygon"]
        rs.recenter_in_TS(centers[b])
        rs.translate_in_TS(shifts[b])
        rp.scale_box(scales[b])
        rp.scale_box(scales[b] * 0.9)
        rp.set_safety_margin(margins[b])
        rp.rotate_in_TS(rot[b])
        rs.set_safety_margin(0.002)
        Ao, ho, _, _ = orc.assemble(ref, given[b].reshape(-1, 1))
        assert_close(G[b], Ao, RTOL_TIGHT, "G of instance %d" % b)
        assert_close(h[b], ho.ravel(), RTOL_TIGHT, "h of instance %d" % b)


def test_csc_hand_off(gpu_api):
    """f3 (biped_mpc_loop.py:57-58): the data arrays of csc_matrix(Q), csc_matrix(A) for a
    whole batch on one structural pattern."""
    import scipy.sparse as sp
    from mpcasm import engine

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    batch = 130
    rng = np.random.default_rng(8)
    given = rng.normal(0, 0.1, [batch, form.given_len])
    asm = engine.Assembler(form, batch=batch)
    P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
    for which, dense in (("P", P), ("G", G)):
        for upper in ((False, True) if which == "P" else (False,)):
            indptr, indices = asm.csc_pattern(which, upper=upper)
            data = asm.export_csc(which, upper=upper).cpu().numpy()
            assert data.shape == (batch, indptr[-1])
            for b in (0, 64, batch - 1):
                mine = sp.csc_matrix((data[b], indices, indptr), shape=dense[b].shape).toarray()
                ref = np.triu(dense[b]) if upper else dense[b]
                assert np.array_equal(mine, ref)
    ref = sp.csc_matrix(P[5])
    indptr, indices = asm.csc_pattern("P")
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(indices, ref.indices)
    assert np.array_equal(asm.export_csc("P").cpu().numpy()[5], ref.data)
    # explicit dense input, fewer instances
    sub = asm.export_csc("G", dense=asm.assemble(given)[2], count=7).cpu().numpy()
    assert sub.shape[0] == 7


def test_lipm3d_c3(gpu_api):
    """C3: N=32, 3 axes, 96 unknowns, 196 inequality rows; golden at B=1, oracle
    on sampled instances of a batch."""
    g = golden("g6_configs")
    form = problems.lipm3d(gpu_api, N=32)
    A, h, Q, q = check_drop_in(form, g, "lipm3d_N32/", parts=False)
    assert Q.shape == (96, 96) and A.shape == (196, 96)
    asm, given, _, (P, qq, G, hh) = batched_case(form, 512, 20261, lambda *a: None)
    for b in (0, 100, 511):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
        assert_close(P[b], Qo, RTOL_TIGHT)
        assert_close(qq[b], qo.ravel(), RTOL_TIGHT)
        assert_close(G[b], Ao, RTOL_TIGHT)
        assert_close(hh[b], ho.ravel(), RTOL_TIGHT)


def test_random_lti_c4(gpu_api):
    """C4 shape (nx=12, nu=6): golden at N=8, the full N=64 size (384 unknowns,
    1536 rows) against the oracle."""
    g = golden("g6_configs")
    form = problems.random_lti(gpu_api, np.random.default_rng(20262), nx=12, nu=6, N=8)
    check_drop_in(form, g, "lti_nx12_nu6_N8/", parts=False)
    form = problems.random_lti(gpu_api, np.random.default_rng(20262), nx=12, nu=6, N=64)
    assert form.optim_len == 384
    asm, given, _, (P, q, G, h) = batched_case(form, 4, 1, lambda *a: None)
    assert G.shape == (4, 1536, 384)
    for b in (0, 3):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
        assert_close(P[b], Qo, RTOL_TIGHT)
        assert_close(q[b], qo.ravel(), RTOL_TIGHT)
        assert_close(G[b], Ao, RTOL_TIGHT)
        assert_close(h[b], ho.ravel(), RTOL_TIGHT)


def test_l_matrices_and_per_row_fields(gpu_api):
    """Every structural branch at once (L on costs and constraints, schedules,
    1-D / 2-D definition coefficients, cross terms, state-space box)."""
    api = gpu_api
    rng = np.random.default_rng(8)
    N = 6
    A, B = problems.random_lti_matrices(rng, 3, 2)
    ext = api.ExtendedSystem.from_cotrol_system(
        api.ControlSystem(["u0", "u1"], ["p", "v", "a"], A, B, axes=["_x", "_y"]), "x", N)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    form.incorporate_definitions({
        "mix_x": api.LineCombo({"p_x": rng.standard_normal((4, N)),
                                "v_x": rng.standard_normal((4, N))}),
        "mix_y": api.LineCombo({"p_y": rng.standard_normal((4, N)), "v_y": 2.0 * np.eye(4, N)}),
        "avg_x": api.LineCombo({"p_x": np.ones(N) / N}),
    })
    form.incorporate_constraint("rows", [
        api.Constraint("mix", rng.uniform(1, 2, 4), axes=["_x", "_y"],
                       arrow=rng.standard_normal((4, 2)), center=rng.standard_normal((4, 2))),
        api.Constraint("p", 3.0, axes=["_x", "_y"], arrow=[1.0, -0.5],
                       L=[rng.standard_normal((3, 2)), rng.standard_normal((3, 2))],
                       schedule=range(2, 4)),
        api.Constraint("avg_x", 1.5),
    ])
    form.incorporate_box("ss", api.Box.state_space(
        "v_x", np.array([[0.0, 1], [1, 0.5], [0.2, -1], [-1, 0]]), schedule=range(0, 2)))
    form.incorporate_goal("g1", api.Cost("mix", 2.0, aim=[0.3, -0.2], axes=["_x", "_y"],
                                         L=rng.standard_normal((2, 4))))
    form.incorporate_goal("g2", api.Cost("a", 0.5, aim=[1.0], axes=["_x"], schedule=range(1, 5),
                                         cross="v", cross_aim=[0.25]))
    form.incorporate_goal("g3", api.Cost("avg_x", 4.0, aim=0.1))
    form.identify_qp_domain(["u0_x", "u1_x", "u0_y", "u1_y"])
    form.make_preview_matrices()
    given = rng.standard_normal([form.given_len, 1])
    A_, h_, Q_, q_ = form.generate_all_qp_matrices(given)
    PMo = orc.preview_matrices(form)
    Ao, ho, Qo, qo = orc.assemble(form, given, PMo)
    for mine, ref in ((A_, Ao), (h_, ho), (Q_, Qo), (q_, qo)):
        assert_close(mine, ref, RTOL_TIGHT)
    for var in form.definitions:
        assert_close(form.PM[var][0], PMo[var][0], RTOL_TIGHT)
        assert_close(form.PM[var][1], PMo[var][1], RTOL_TIGHT)


def test_errors_and_edges(gpu_api):
    from mpcasm import capi
    from mpcasm.engine import Assembler

    form = problems.body_case(gpu_api)
    asm = Assembler(form, batch=3)
    with pytest.raises(ValueError):
        asm.assemble(np.zeros([2, form.given_len]))              # wrong batch
    with pytest.raises(KeyError):
        asm.set_param("cost", "nope", "aim", [[0.0]])
    with pytest.raises(KeyError):
        asm.bind_source(("LIP", 9), np.zeros(3))
    with pytest.raises(ValueError):
        asm.bind_source(("LIP", 0), np.zeros((2, 2, 2)))
    lib = capi.load()
    assert lib.mpcasm_assemble(None, None, None, None, None, None, None, None, None, None,
                               1, None) == -1
    # P without q is refused
    P, q, G, h = asm.assemble(np.zeros([3, form.given_len]))
    ptrs, strides = asm._src_args()
    rc = lib.mpcasm_assemble(asm._handle, ptrs, strides, asm.params.data_ptr(), P.data_ptr(),
                             P.data_ptr(), None, G.data_ptr(), h.data_ptr(),
                             asm._work.data_ptr(), 3, None)
    assert rc == -1
    # a definition that depends on an undefined variable is refused at incorporation
    with pytest.raises(AssertionError):
        form.incorporate_definition("bad", gpu_api.LineCombo({"nope_x": 1}))


def test_edge_shapes(gpu_api):
    """Degenerate shapes: costs only (no inequality rows), constraints only, every
    domain variable unknown (empty given vector), batches of one and zero."""
    import torch
    from mpcasm.engine import Assembler

    api = gpu_api
    rng = np.random.default_rng(12)
    A, B = problems.random_lti_matrices(rng, 3, 1)
    ext = api.ExtendedSystem.from_cotrol_system(
        api.ControlSystem(["u"], ["p", "v", "a"], A, B), "x", 5)

    def base():
        form = api.Formulation()
        form.incorporate_dynamics("plant", ext)
        return form

    # costs only
    form = base()
    form.incorporate_goal("track", api.Cost("p", 2.0, aim=[0.5]))
    form.identify_qp_domain(["u"])
    form.make_preview_matrices()
    given = rng.standard_normal([form.given_len, 1])
    Q, q = form.generate_all_qp_costs(given)
    PM = orc.preview_matrices(form)
    Qo, qo = orc.qp_all_costs(form, PM, given)
    assert_close(Q, Qo, RTOL_TIGHT)
    assert_close(q, qo, RTOL_TIGHT)
    asm = Assembler(form, batch=2)
    P, qq, G, h = asm.assemble(np.tile(given.T, (2, 1)))
    assert G is None and h is None and P.shape == (2, 5, 5)

    # constraints only, one instance, odd number of unknowns (scalar store paths)
    form = base()
    form.incorporate_constraint("cap", [api.Constraint("v", 1.0), api.Constraint("p", 2.0, arrow=[-1])])
    form.identify_qp_domain(["u"])
    form.make_preview_matrices()
    Ag, hg = form.generate_all_qp_constraints(given)
    Ao, ho = orc.qp_all_constraints(form, orc.preview_matrices(form), given)
    assert_close(Ag, Ao, RTOL_TIGHT)
    assert_close(hg, ho, RTOL_TIGHT)

    # every domain variable unknown: the given vector is empty
    form = base()
    form.incorporate_goal("track", api.Cost("p", 1.0, aim=[0.1]))
    form.incorporate_constraint("cap", api.Constraint("v", 1.0))
    form.identify_qp_domain(["u", "x0"])
    form.make_preview_matrices()
    assert form.given_len == 0 and form.optim_len == 8
    empty = form.arrange_given({})
    A_, h_, Q_, q_ = form.generate_all_qp_matrices(empty)
    Ao, ho, Qo, qo = orc.assemble(form, np.zeros([0, 1]))
    for mine, ref in ((A_, Ao), (h_, ho), (Q_, Qo), (q_, qo)):
        assert_close(mine, ref, RTOL_TIGHT)

    # a batch of zero instances is a no-op
    asm = Assembler(problems.body_case(api), batch=0)
    P, qq, G, h = asm.assemble(torch.zeros((0, asm.ng), dtype=torch.float64, device="cuda"))
    assert P.shape[0] == 0 and G.shape[0] == 0


def test_wide_hessian_with_a_crossed_term(gpu_api):
    """no >= 128 takes the LDS-tiled GEMM for P; a crossed cost makes P non-symmetric, so
    every block (not only bi <= bj) has to be computed."""
    form = problems.random_lti(gpu_api, np.random.default_rng(7), nx=4, nu=3, N=48)
    form.incorporate_goal("crossed", gpu_api.Cost("s0", 0.3, aim=[0.1], cross="s2",
                                                  cross_aim=[-0.2]))
    form.make_preview_matrices()
    assert form.optim_len == 144
    given = np.random.default_rng(8).standard_normal([form.given_len, 1])
    A, h, Q, q = form.generate_all_qp_matrices(given)
    Ao, ho, Qo, qo = orc.assemble(form, given)
    assert np.abs(Qo - Qo.T).max() > 1e-6
    for mine, ref in ((A, Ao), (h, ho), (Q, Qo), (q, qo)):
        assert_close(mine, ref, RTOL_TIGHT)


def test_ticks_reuse_compiled_plans(gpu_api, kernel_path, monkeypatch):
    """The per-tick API compiles a plan once per *structure*: 18 ticks of the walking loop see
    three structures (two steps in the preview / one / the step just taken), every other tick
    re-uploads horizon matrices and parameters of a cached plan -- and the per-cost / per-limit
    calls reuse theirs.  Results as in test_biped_ticks_drop_in (checked there tick by tick);
    here: the number of compilations, and that an in-place edit of an L matrix or of a
    definition's coefficients is seen (content, not identity, keys the cache)."""
    if kernel_path not in RESIDENT[:1]:
        pytest.skip("host-side caching: one path is enough")
    import mpcasm.engine as engine

    compiled = []
    real = engine.compile_plan
    monkeypatch.setattr(engine, "compile_plan", lambda *a, **k: compiled.append(1) or real(*a, **k))
    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    clock = problems.StepClock(conf.step_samples, conf.num_steps)
    rng = np.random.default_rng(5)
    for tick in range(18):
        form.update(step_times=clock.step_times, step_count=clock.step_count)
        given = form.arrange_given(problems.biped_given_collector(form, rng, 0.01))
        A, h, Q, q = form.generate_all_qp_matrices(given)
        Ao, ho, Qo, qo = orc.assemble(form, given)
        assert_close(A, Ao, RTOL_TIGHT), assert_close(h, ho, RTOL_TIGHT)
        assert_close(Q, Qo, RTOL_TIGHT), assert_close(q, qo, RTOL_TIGHT)
        Q1, q1 = form.generate_qp_cost(form.goals["track vel_x"], given)
        maps = orc.qp_index_maps(form.domain, form.optim_variables)
        PMo = orc.preview_matrices(form, maps)
        Qo1, qo1 = orc.qp_cost(PMo, form.goals["track vel_x"], given)
        assert_close(Q1, Qo1, RTOL_TIGHT), assert_close(q1, qo1, RTOL_TIGHT)
        clock.tick()
    assert len(compiled) <= 8, len(compiled)      # 18 ticks x 2 assemblers without the cache

    # content keys the cache: an in-place edit of a derived definition's coefficient
    form2 = problems.body_case(gpu_api)
    given = np.arange(form2.given_len, dtype=float).reshape(-1, 1)
    before = form2.generate_all_qp_matrices(given)
    form2.definitions["DCM_x"].matrices[1] = 0.5
    form2.make_preview_matrices()
    after = form2.generate_all_qp_matrices(given)
    want = orc.assemble(form2, given)
    for mine, ref in zip(after, want):
        assert_close(mine, ref, RTOL_TIGHT)
    assert not np.allclose(before[2], after[2])
    # ... and of an L matrix, in place
    lim = form2.constraints["steppingArea"][0]
    lim.L[0][:] = 2.0 * np.eye(9)
    mine = form2.generate_qp_constraint(lim, given)
    ref = orc.qp_constraint(orc.preview_matrices(
        form2, orc.qp_index_maps(form2.domain, form2.optim_variables)), lim, given)
    assert_close(mine[0], ref[0], RTOL_TIGHT), assert_close(mine[1], ref[1], RTOL_TIGHT)


def test_batched_state_space_box_transforms(gpu_api, kernel_path):
    """f4, second half (restrictions.py:342-378, 390-404, 417-431): a box made by
    Box.state_space (facet normals as L), re-centred and translated IN THE STATE SPACE for every
    instance at once (mpcasm_box_transform_ss), then scaled and shrunk by the task-space
    operations -- against the same calls on the host Box, instance by instance, assembled by the
    oracle."""
    import copy

    from mpcasm import engine
    from mpcasm.boxes import BoxBatch

    if kernel_path not in RESIDENT:
        pytest.skip("the transforms act on the parameters, one assembly path is enough")
    form = problems.body_case(gpu_api)
    verts = np.array([[0.3, 0.1], [-0.2, 0.25], [-0.3, -0.15], [0.2, -0.3]])
    form.incorporate_box("dcm window", gpu_api.Box.state_space("DCM_x", verts, schedule=range(7, 9)))
    form.make_preview_matrices()
    batch = 7
    rng = np.random.default_rng(33)
    given = rng.normal(0, 0.3, [batch, form.given_len])
    asm = engine.Assembler(form, batch=batch)
    window = BoxBatch(asm, form, "dcm window")
    assert window.ss_dim == 2 and window.lrows == 1 and window.axes == 1
    centers = rng.normal(0, 0.2, [batch, 2, 1])
    shifts = rng.normal(0, 0.05, [batch, 2, 1])
    window.recenter_in_SS(centers)
    window.translate_in_SS(shifts)
    window.translate_in_SS(np.array([[0.01], [-0.02]]))      # one point for all
    window.scale_box(rng.uniform(0.7, 1.2, batch) * 0 + 1.1)
    window.set_safety_margin(0.01)
    G, h = (t.cpu().numpy() for t in asm.assemble(given)[2:])
    for b in (0, 3, batch - 1):
        ref = copy.deepcopy(form)
        box = ref.constraint_boxes["dcm window"]
        box.recenter_in_SS(centers[b])
        box.translate_in_SS(shifts[b])
        box.translate_in_SS(np.array([[0.01], [-0.02]]))
        box.scale_box(1.1)
        box.set_safety_margin(0.01)
        Ao, ho, _, _ = orc.assemble(ref, given[b].reshape(-1, 1))
        assert_close(G[b], Ao, RTOL_TIGHT, "G of instance %d" % b)
        assert_close(h[b], ho.ravel(), RTOL_TIGHT, "h of instance %d" % b)
    # misuse: the reference's shape error, and a task-space box has no state space
    with pytest.raises(ValueError, match="must have 2 rows and 1 columns"):
        window.recenter_in_SS(np.zeros((batch, 3, 1)))
    with pytest.raises(ValueError, match="task space only"):
        BoxBatch(asm, form, "kine").recenter_in_SS(np.zeros((2, 2)))


def test_csc_data_written_by_the_assembly(gpu_api, kernel_path):
    """f3 without a second pass: an Assembler built with csc=... gets the ``data`` arrays of
    ``scipy.sparse.csc_matrix(Q)`` (upper triangle, what OSQP takes) and ``csc_matrix(A)``
    (biped_mpc_loop.py:57-58) from the assembly kernel itself.  Against scipy on the oracle's
    dense matrices, instance by instance, and against the dense assembly of the same inputs
    entry for entry (bit for bit: the arithmetic is the same); every stored entry is written
    (buffers pre-set to NaN).  The biped with its horizon matrices built on chip, and the
    reference's test_body problem (crossed cost: P stored in full)."""
    import scipy.sparse as sp
    import torch
    from mpcasm import engine

    if kernel_path not in RESIDENT:
        pytest.skip("only the persistent kernel writes the CSC form")
    biped = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    biped.update(step_times=np.array([6, 14]), step_count=0)
    for form, lti, kind, batch in ((biped, ["LIP"], "upper", 301), (problems.body_case(gpu_api), [], "full", 9)):
        rng = np.random.default_rng(41)
        given = rng.normal(0, 0.2, [batch, form.given_len])
        sparse = engine.Assembler(form, batch=batch, lti=lti, csc=kind)
        dense = engine.Assembler(form, batch=batch, lti=lti)
        c = sparse.csc
        assert sparse.csc_pattern("P") == c["P"] and sparse.csc_pattern("G") == c["G"]
        out = (torch.full((batch, c["pnnz"]), float("nan"), dtype=torch.float64, device="cuda"),
               torch.full((batch, form.optim_len), float("nan"), dtype=torch.float64, device="cuda"),
               torch.full((batch, c["gnnz"]), float("nan"), dtype=torch.float64, device="cuda"),
               torch.full((batch, dense.nc), float("nan"), dtype=torch.float64, device="cuda"))
        Pd, q, Gd, h = (t.cpu().numpy() for t in sparse.assemble(given, out=out))
        P, q2, G, h2 = (t.cpu().numpy() for t in dense.assemble(given))
        assert not any(np.isnan(t).any() for t in (Pd, q, Gd, h))
        assert np.array_equal(Pd, P.reshape(batch, -1)[:, c["p_flat"]])
        assert np.array_equal(Gd, G.reshape(batch, -1)[:, c["g_flat"]])
        assert np.array_equal(q, q2) and np.array_equal(h, h2)
        for b in (0, batch // 2, batch - 1):
            Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
            Qs = sp.csc_matrix(np.triu(Qo) if kind == "upper" else Qo)
            As = sp.csc_matrix(Ao)
            mine_P = sp.csc_matrix((Pd[b], c["P"][1], c["P"][0]), shape=Qo.shape)
            mine_G = sp.csc_matrix((Gd[b], c["G"][1], c["G"][0]), shape=Ao.shape)
            assert_close(mine_P.toarray(), Qs.toarray(), RTOL_TIGHT, "P of instance %d" % b)
            assert_close(mine_G.toarray(), As.toarray(), RTOL_TIGHT, "G of instance %d" % b)
            # scipy stores no entry the pattern lacks
            assert not (Qs.toarray() != 0)[~sparse.plan.P_pattern].any()
            assert not (As.toarray() != 0)[~sparse.plan.G_pattern].any()
        with pytest.raises(ValueError, match="writes the CSC form itself"):
            sparse.export_csc("P")


def test_options_of_one_plan(gpu_api, kernel_path):
    """mpcasm_plan_set_option: a plan's own kernel path / per-plan compilation, independent of the
    process-wide hooks the fixture sets -- two assemblers of the same problem, one pinned to the
    staged pipeline and one to the per-plan persistent kernel, give the same QPs whatever the
    process-wide choice is."""
    from mpcasm import capi, engine

    if kernel_path not in ("resident", "staged"):
        pytest.skip("two process-wide settings are enough")
    form = problems.body_case(gpu_api)
    batch = 40
    given = np.random.default_rng(8).normal(0, 0.3, [batch, form.given_len])
    staged = engine.Assembler(form, batch=batch)
    staged.set_option(capi.OPT_PATH, 2)
    per_plan = engine.Assembler(form, batch=batch)
    per_plan.set_option(capi.OPT_PATH, 0)
    per_plan.set_option(capi.OPT_JIT, 1)
    a = [t.cpu().numpy() for t in staged.assemble(given)]
    b = [t.cpu().numpy() for t in per_plan.assemble(given)]
    for x, y, name in zip(a, b, "PqGh"):
        assert_close(x, y, 1e-13, name)
    Ao, ho, Qo, qo = orc.assemble(form, given[7].reshape(-1, 1))
    assert_close(b[0][7], Qo, RTOL_TIGHT, "P"), assert_close(b[2][7], Ao, RTOL_TIGHT, "G")
    with pytest.raises(capi.MpcasmError):
        staged.set_option(capi.OPT_P_DIRECT, 1)          # read at plan creation: not a plan option
    staged.set_option(capi.OPT_PATH, -1)                 # back to the process-wide value
