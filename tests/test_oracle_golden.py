"""CPU: the oracle and this repository's host classes against the golden vectors
captured from the real reference (tests/golden/make_golden.py).

Every problem is rebuilt with the repository's own ``mpc_interface`` API (host
logic; ``extend_matrices`` served by the oracle through the ``cpu_api`` fixture),
then the oracle runs on those objects and must reproduce the reference's numbers:
index maps bit-exact, floating point within 1e-10 relative.
"""
import json

import numpy as np
import pytest

from helpers import RTOL, assert_close, golden, ranges_from_json
from mpcasm import problems
from oracle import qp_oracle as orc


# --------------------------------------------------------------------------- G1
def test_extend_matrices_reference_fixture():
    """The reference's own known-answer test (test_dynamics.py:139-152): J->CCC,
    tau=0.1, omega=3.3445, N=36 against python/tests/LIP_matrices."""
    g = golden("g1_extend")
    S, U = orc.extend_matrices(36, g["lip36/A"], g["lip36/B"])
    assert S.shape == (36, 3, 3) and len(U) == 1 and U[0].shape == (36, 36, 3)
    assert_close(S, g["lip36/S"], what="S")
    assert_close(U[0], g["lip36/U0"], what="U")


@pytest.mark.parametrize("n,m,N", [(3, 1, 16), (3, 1, 32), (3, 1, 100), (8, 6, 20),
                                   (12, 6, 64), (1, 1, 5), (2, 3, 1)])
def test_extend_matrices_random_lti(n, m, N):
    g = golden("g1_extend")
    key = "lti_n%d_m%d_N%d/" % (n, m, N)
    A, B = g[key + "A"], g[key + "B"]
    S, U = orc.extend_matrices(N, A, B)
    # shape contract of the reference's test_tools.py:19-33
    assert S.shape == (N, n, n) and isinstance(U, list) and len(U) == m
    assert U[0].shape == (N, N, n)
    assert_close(S, g[key + "S"], what="S")
    if key + "U" in g:
        assert_close(np.stack(U), g[key + "U"], what="U")
    else:
        rows = g[key + "U_rows"]
        assert_close(U[0][rows], g[key + "U_first"], what="U first")
        assert_close(U[m - 1][rows], g[key + "U_last"], what="U last")
        assert_close(np.array([u.sum() for u in U]), g[key + "U_sums"], what="U sums")
    # LTV generalisation degenerates to the LTI result
    Sl, Ul = orc.extend_matrices_ltv(N, np.stack([A] * N), np.stack([B] * N))
    assert_close(Sl, S, 1e-13, "ltv S")
    assert_close(np.stack(Ul), np.stack(U), 1e-13, "ltv U")


def test_extend_matrices_structure():
    """Layout identities of SURVEY.md section 3.3: block-lower-triangular Toeplitz."""
    rng = np.random.default_rng(5)
    A, B = rng.standard_normal((4, 4)) / 2, rng.standard_normal((4, 2))
    S, U = orc.extend_matrices(7, A, B)
    for k in range(7):
        assert_close(S[k], np.linalg.matrix_power(A, k + 1).T, 1e-12)
        for l in range(7):
            for j in range(2):
                want = (np.linalg.matrix_power(A, k - l) @ B)[:, j] if l <= k else np.zeros(4)
                assert_close(U[j][k, l], want, 1e-12)


@pytest.fixture(scope="module")
def c_oracle():
    """The compiled twin (oracle/extend_matrices.c); built here when it is not yet."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)
    from oracle import c_oracle as c

    c.load()
    return c


def test_c_restatement_against_the_reference_vectors(c_oracle):
    """oracle/extend_matrices.c: the reference's known-answer fixture, the seeded random
    systems, the numpy oracle (same recurrence; BLAS may sum in another order),
    and the LTV variant against its numpy counterpart."""
    g = golden("g1_extend")
    S, U = c_oracle.extend_matrices(36, g["lip36/A"], g["lip36/B"])
    assert_close(S, g["lip36/S"], what="S")
    assert_close(U[0], g["lip36/U0"], what="U")
    for n, m, N in [(3, 1, 16), (8, 6, 20), (12, 6, 64), (1, 1, 5), (2, 3, 1)]:
        key = "lti_n%d_m%d_N%d/" % (n, m, N)
        A, B = g[key + "A"], g[key + "B"]
        S, U = c_oracle.extend_matrices(N, A, B)
        So, Uo = orc.extend_matrices(N, A, B)
        assert_close(S, g[key + "S"], what="S")
        assert_close(S, So, 1e-13)
        assert_close(np.stack(U), np.stack(Uo), 1e-13)
    rng = np.random.default_rng(9)
    A = rng.standard_normal((3, 7, 4, 4)) / 2
    B = rng.standard_normal((3, 7, 4, 2))
    S, U = c_oracle.extend_matrices_batch(A, B, 7, ltv=True)
    for b in range(3):
        So, Uo = orc.extend_matrices_ltv(7, A[b], B[b])
        assert_close(S[b], So, 1e-14)
        assert_close(U[b], np.stack(Uo), 1e-14)


# ---------------------------------------------------------------------- G2 .. G6
def check_snapshot(form, g, prefix):
    """Index maps bit-exact; PM, per-part blocks and stacked blocks within RTOL."""
    maps = orc.qp_index_maps(form.domain, form.optim_variables)
    assert {k: form.optim_ID[k] for k in form.optim_variables} == \
        ranges_from_json(g[prefix + "optim_ID"])
    assert {k: form.given_ID[k] for k in form.given_variables} == \
        ranges_from_json(g[prefix + "given_ID"])
    assert list(maps["optim_ID"].items()) == list(ranges_from_json(g[prefix + "optim_ID"]).items())
    assert list(maps["given_ID"].items()) == list(ranges_from_json(g[prefix + "given_ID"]).items())
    assert list(form.definitions.keys()) == json.loads(str(g[prefix + "definitions"]))

    given = g[prefix + "given"]
    PM = orc.preview_matrices(form, maps)
    for var in form.definitions:
        if prefix + "PM/" + var + "/Mg" in g:
            assert_close(PM[var][0], g[prefix + "PM/" + var + "/Mg"], what=var + " Mg")
            assert_close(PM[var][1], g[prefix + "PM/" + var + "/Mo"], what=var + " Mo")
    for k, limit in enumerate(orc.all_limits(form)):
        if prefix + "limit%d/A" % k in g:
            A, h = orc.qp_constraint(PM, limit, given)
            assert_close(A, g[prefix + "limit%d/A" % k], what="limit A")
            assert_close(h, g[prefix + "limit%d/h" % k], what="limit h")
    for name, cost in form.goals.items():
        if prefix + "cost/" + name + "/Q" in g:
            Q, q = orc.qp_cost(PM, cost, given)
            assert_close(Q, g[prefix + "cost/" + name + "/Q"], what="cost Q")
            assert_close(q, g[prefix + "cost/" + name + "/q"], what="cost q")
    A, h, Q, q = orc.assemble(form, given, PM, maps)
    for mine, nm in ((A, "A"), (h, "h"), (Q, "Q"), (q, "q")):
        assert_close(mine, g[prefix + nm], RTOL, prefix + nm)
    # the compiled port of the same path (oracle/assemble_port.c) against the same vectors of the reference
    port = _c_port()
    Gc, hc, Pc, qc = port.Recipe(form).assemble(None, None, np.asarray(given).reshape(1, -1))
    for mine, nm in ((Gc[0], "A"), (hc[0][:, None], "h"), (Pc[0], "Q"), (qc[0][:, None], "q")):
        assert_close(mine, g[prefix + nm], RTOL, prefix + nm + " (C port)")
    return A, h, Q, q


def _c_port():
    """oracle/c_port.py over oracle/liboracle.so; built here when it is not yet (or is an older one)."""
    import os
    import subprocess

    from oracle import c_port
    try:
        c_port.load()
    except (RuntimeError, AttributeError):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call(["make", "-B", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL)
        c_port._lib = None
        c_port.load()
    return c_port


def test_c_port_with_a_system_and_constants_of_its_own_per_instance(cpu_api):
    """The compiled port on the bench's workload shape: every instance its own (A, B) (horizon matrices
    extended per instance, tools.py:14-33; where they land in the preview matrices found by probing the
    definitions), its own `given` and its own aim of the velocity cost -- against the numpy oracle run
    instance by instance the way the reference re-plans (dynamics.py:222-231 + body.py:142-348)."""
    port = _c_port()
    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(cpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)
    N, B = conf.horizon_lenght, 6
    rng = np.random.default_rng(21)
    lip = form.dynamics["LIP"]
    A0, B0 = np.array([[1, 0.1, 0.005], [0, 1, 0.1], [0, 0, 1]]), np.array([[0.1 ** 3 / 6], [0.005], [0.1]])
    As = A0[None] + rng.normal(0, 0.01, [B, 3, 3])
    Bs = B0[None] + rng.normal(0, 0.01, [B, 3, 1])
    given = rng.normal(0, 0.2, [B, form.given_len])
    aims = rng.normal(0.3, 0.1, [B, 1])
    recipe = port.Recipe(form, per_instance="LIP")
    consts = np.tile(recipe.consts, (B, 1))
    for field in ("aim", "cross_aim"):
        consts[:, recipe.const_slice("cost", "track vel_x", field)] = aims
    G, h, P, q = recipe.assemble(As, Bs, given, consts)
    saved, aim0 = list(lip.matrices), np.array(form.goals["track vel_x"].aim)
    try:
        for b in range(B):
            S, U = orc.extend_matrices(N, As[b], Bs[b])
            lip.matrices = U + [S]
            lip.update_definitions()
            form.goals["track vel_x"].update(aim=aims[b])
            Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
            assert_close(G[b], Ao, 1e-13, "G"), assert_close(h[b], ho.ravel(), 1e-13, "h")
            assert_close(P[b], Qo, 1e-13, "P"), assert_close(q[b], qo.ravel(), 1e-13, "q")
    finally:
        lip.matrices = saved
        lip.update_definitions()
        form.goals["track vel_x"].update(aim=aim0)
    # the formulation's own matrices are back where the probing found them
    assert all(np.array_equal(a, b) for a, b in zip(lip.matrices, saved))


def test_body_case(cpu_api):
    g = golden("g2_body")
    form = problems.body_case(cpu_api)
    A, h, Q, q = check_snapshot(form, g, "arange/")
    # shapes the reference's test_body.py:140-162 asserts
    assert A.shape == (54, form.optim_len) and h.shape == (54, 1)
    assert Q.shape == (form.optim_len,) * 2 and q.shape == (form.optim_len, 1)
    check_snapshot(form, g, "random/")
    # crossed cost gives a non-symmetric Hessian (SURVEY.md section 8a quirk i)
    assert np.abs(Q - Q.T).max() > 1e-6


@pytest.mark.parametrize("step_samples", [8, 12])
def test_biped_ticks(cpu_api, step_samples):
    """Walking ticks: the QP width alternates (34/36 at N=16, 50/52 at N=24)."""
    conf = problems.BipedConfig(step_samples=step_samples)
    g = golden("g3_biped_N%d" % conf.horizon_lenght)
    form = problems.biped(cpu_api, conf)
    clock = problems.StepClock(conf.step_samples, form.domain["Ds_x"])
    keep = set(int(t) for t in g["ticks"])
    shapes = []
    for tick in range(18):
        form.update(step_times=clock.step_times, step_count=clock.step_count)
        if tick in keep:
            p = "tick%02d/" % tick
            assert np.array_equal(clock.step_times, g[p + "step_times"])
            assert clock.step_count == int(g[p + "step_count"])
            box = form.constraint_boxes["stepping area"]
            assert_close(np.stack([l.center for l in box.constraints]),
                         g[p + "stepping_centers"], 1e-15, "centers")
            check_snapshot(form, g, p)
        nc = orc.assemble(form, np.zeros([form.given_len, 1]))[0].shape[0]
        shapes.append([tick, clock.step_count, form.optim_len, nc])
        clock.tick()
    assert np.array_equal(np.array(shapes), g["shapes"])


def test_lipm3d_and_lti_configs(cpu_api):
    g = golden("g6_configs")
    form = problems.lipm3d(cpu_api, N=32)
    A, h, Q, q = check_snapshot(form, g, "lipm3d_N32/")
    assert Q.shape == (96, 96) and A.shape == (196, 96)
    form = problems.random_lti(cpu_api, np.random.default_rng(20262), nx=12, nu=6, N=8)
    A, h, Q, q = check_snapshot(form, g, "lti_nx12_nu6_N8/")
    assert Q.shape == (48, 48) and A.shape == (2 * 12 * 8, 48)
