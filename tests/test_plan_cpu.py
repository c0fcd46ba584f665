"""CPU: the host plan compiler (mpcasm.plan) against the oracle.

The plan tables are executed by tests/plan_emulator.py, a numpy interpreter of
csrc/plan_tables.h, so that the compiler's flattening of the definition graph,
row-set bookkeeping, gterm / limit records and parameter layout are verified
without a GPU.  The same tables drive the HIP kernels (tests/test_gpu_*.py).
"""
import ctypes

import numpy as np
import pytest

import plan_emulator
from helpers import assert_close, lti_tracking_problem
from mpcasm import capi, problems
from mpcasm.plan import _H, compile_plan, structure_fingerprint
from oracle import qp_oracle as orc


def check_plan(form, given, tol=1e-12):
    plan = compile_plan(form)
    out = plan_emulator.run(plan, given)
    PM = orc.preview_matrices(form)
    A, h, Q, q = orc.assemble(form, given, PM)
    assert (plan.ng, plan.no, plan.nc) == (form.given_len, form.optim_len, A.shape[0])
    assert_close(out["P"], Q, tol, "P")
    assert_close(out["q"], q.ravel(), tol, "q")
    assert_close(out["G"], A, tol, "G")
    assert_close(out["h"], h.ravel(), tol, "h")
    # the fused program (per-element op lists) builds the same workspace
    assert_close(plan_emulator.run_fused_workspace(plan, given), out["V"], tol, "fused V")
    for var, (r0, rows) in plan.pm_rows.items():
        block = out["PM"][r0:r0 + rows]
        assert_close(block[:, :plan.ng], PM[var][0], tol, var + " Mg")
        assert_close(block[:, plan.ng:], PM[var][1], tol, var + " Mo")
    # the column tables / stage list of the tiled kernel (built for every plan)
    assert plan.itab[_H["T_CI_OK"]] == 1
    # ... unrolled per base row for f2: the rows of every definition from [given ; optim]
    optim = np.random.default_rng(9).standard_normal(plan.no)
    rows = plan_emulator.run_preview_tables(plan, given, optim)
    for var, (r0, nrows) in plan.pm_rows.items():
        Mg, Mo = PM[var]
        assert_close(rows[r0:r0 + nrows], (Mg @ np.asarray(given).reshape(-1, 1)).ravel() + Mo @ optim, tol,
                     var + " preview rows")
    til = plan_emulator.run_tiled(plan, given)
    for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
        assert_close(til[key], ref, tol, "tiled " + key)
    # the tables of the persistent kernel, when the problem fits it
    if plan.itab[_H["RS_OK"]]:
        res = plan_emulator.run_resident(plan, given)
        for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
            assert_close(res[key], ref, tol, "resident " + key)
        # ... and with every row-set keeping only its own columns (where the plan's paths of G allow)
        small = compile_plan(form, workspace="compact")
        if small.workspace.compact:
            assert small.itab[_H["RS_OK"]] and small.workspace.doubles <= plan.workspace.doubles
            res = plan_emulator.run_resident(small, given)
            for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
                assert_close(res[key], ref, tol, "compact " + key)
    return plan


def test_csc_plans_write_what_scipy_would_store(cpu_api):
    """f3, written by the assembly itself: a plan compiled with csc='upper' / 'full' carries, per
    stored entry, where the persistent kernel finds it (P) or how it computes it (G).  The
    emulated kernel's data arrays, put on the plan's (indptr, indices), are the oracle's P (its
    upper triangle) and G -- biped_mpc_loop.py:57-58 -- for the biped, K1 fused and not, and the
    reference's test_body problem (crossed cost: a P that is not symmetric, stored in full)."""
    import scipy.sparse as sp

    biped = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    biped.update(step_times=np.array([6, 14]), step_count=0)
    lib = capi.load()
    for form, lti, kind in ((biped, (), "upper"), (biped, ("LIP",), "upper"),
                            (problems.body_case(cpu_api), (), "full")):
        rng = np.random.default_rng(5)
        given = rng.standard_normal([form.given_len, 1])
        plan = compile_plan(form, lti=lti, csc=kind)
        A, h, Q, q = orc.assemble(form, given)
        srcs = [s.array for s in plan.sources]
        for g in plan.lti:                      # (A, B) travel in the group's first two sources
            mats = form.dynamics[g["name"]].matrices
            srcs[g["ids"][0]] = mats[-1][0].T.copy()
            srcs[g["ids"][1]] = np.stack([mats[j][0, 0, :] for j in range(g["m"])], axis=1)
        res = plan_emulator.run_resident(plan, given, sources=srcs)
        c = plan.csc
        assert res["P_data"].shape == (c["pnnz"],) and res["G_data"].shape == (c["gnnz"],)
        Pm = sp.csc_matrix((res["P_data"], c["P"][1], c["P"][0]), shape=Q.shape).toarray()
        Gm = sp.csc_matrix((res["G_data"], c["G"][1], c["G"][0]), shape=A.shape).toarray()
        assert_close(Pm, np.triu(Q) if kind == "upper" else Q, 1e-12, "P")
        assert_close(Gm, A, 1e-12, "G")
        assert_close(res["q"], q.ravel(), 1e-12, "q"), assert_close(res["h"], h.ravel(), 1e-12, "h")
        # ... and the library takes the tables (the device step is all that is missing here)
        handle = ctypes.c_void_p()
        rc = lib.mpcasm_plan_create(plan.itab.ctypes.data, plan.itab.size, plan.dtab.ctypes.data,
                                    plan.dtab.size, ctypes.byref(handle))
        assert rc in (0, -4), rc
        if rc == 0:
            lib.mpcasm_plan_destroy(handle)
        # a corrupted entry is refused: a column outside the row, an arrow of another row
        for word, value in ((_H["OFF_CSC_P"], None), (_H["OFF_CSC_G"], None)):
            bad = plan.itab.copy()
            bad[bad[word]] = 1 << 30
            assert lib.mpcasm_plan_create(bad.ctypes.data, bad.size, plan.dtab.ctypes.data,
                                          plan.dtab.size, ctypes.byref(handle)) == -2
    # a problem that does not fit on chip has no such plan
    with pytest.raises(ValueError):
        compile_plan(problems.random_lti(cpu_api, np.random.default_rng(3), N=16), csc="upper")


def test_structural_csc_patterns(cpu_api):
    """f3: the batch-wide sparsity patterns contain every numeric non-zero of the oracle's
    P and G; with generic numbers the P pattern IS scipy's; CSC ordering as scipy's."""
    import scipy.sparse as sp
    from mpcasm.plan import csc_pattern

    cases = [problems.biped(cpu_api, problems.BipedConfig(step_samples=8)),
             problems.body_case(cpu_api)]
    cases[0].update(step_times=np.array([6, 14]), step_count=0)
    for form in cases:
        rng = np.random.default_rng(11)
        given = rng.standard_normal([form.given_len, 1])
        plan = compile_plan(form)
        A, h, Q, q = orc.assemble(form, given)
        assert plan.P_pattern.shape == Q.shape and plan.G_pattern.shape == A.shape
        assert ((Q != 0) <= plan.P_pattern).all() and ((A != 0) <= plan.G_pattern).all()
        for dense, mask in ((Q, plan.P_pattern), (A, plan.G_pattern)):
            indptr, indices, flat = csc_pattern(mask)
            mine = sp.csc_matrix((dense.ravel()[flat], indices, indptr), shape=dense.shape)
            assert np.array_equal(mine.toarray(), dense)
            assert (np.diff(indptr) >= 0).all() and indptr[-1] == mask.sum() == flat.size
            for c in range(dense.shape[1]):                      # rows ascending inside a column
                assert (np.diff(indices[indptr[c]:indptr[c + 1]]) > 0).all()
        up = csc_pattern(plan.P_pattern, upper=True)
        assert (up[2] // Q.shape[1] <= up[2] % Q.shape[1]).all()
    # the biped: U above its diagonal is a known zero, so the ZMP rows do not fill P
    plan = compile_plan(cases[0])
    ref = sp.csc_matrix(orc.assemble(cases[0], np.random.default_rng(1).standard_normal(
        [cases[0].given_len, 1]))[2])
    indptr, indices, _ = csc_pattern(plan.P_pattern)
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(indices, ref.indices)


def test_horizon_matrices_generated_on_chip_plan(cpu_api):
    """``lti=[name]``: the image carries (A, B) and tables built from them instead of S, U;
    compose ops that read U above its diagonal are gone."""
    for times in ([7, 15], [6, 14]):
        form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
        form.update(step_times=np.array(times), step_count=0)
        given = np.random.default_rng(3).standard_normal([form.given_len, 1])
        full = compile_plan(form)
        plan = compile_plan(form, lti=["LIP"])
        assert plan.itab[_H["RS_OK"]] == 1 and plan.itab[_H["RS_NLTI"]] == 1
        assert plan.itab[_H["RS_IMG"]] < full.itab[_H["RS_IMG"]] // 2
        g = plan.lti[0]
        assert (g["n"], g["m"], g["N"]) == (3, 1, 16)
        lip = form.dynamics["LIP"]
        A_sys = lip.matrices[-1][0].T.copy()
        B_sys = lip.matrices[0][0, 0, :].reshape(3, 1).copy()
        Sx, Ux = orc.extend_matrices(16, A_sys, B_sys)
        assert_close(Sx, lip.matrices[-1], 1e-13)
        assert_close(Ux[0], lip.matrices[0], 1e-13)
        srcs = [s.array for s in plan.sources]
        srcs[g["ids"][0]], srcs[g["ids"][1]] = A_sys, B_sys
        res = plan_emulator.run_resident(plan, given, sources=srcs)
        A, h, Q, q = orc.assemble(form, given)
        for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
            assert_close(res[key], ref, 1e-12, key)
    with pytest.raises(KeyError):
        compile_plan(form, lti=["no such dynamics"])
    # several inputs, n > 4, a horizon that is no power of two
    rng = np.random.default_rng(5)
    form = problems.random_lti(cpu_api, rng, nx=5, nu=2, N=5)
    plan = compile_plan(form, lti=["plant"])
    g = plan.lti[0]
    assert (g["n"], g["m"], g["N"]) == (5, 2, 5) and plan.itab[_H["RS_OK"]] == 1
    plant = form.dynamics["plant"]
    A_sys = plant.matrices[-1][0].T.copy()
    B_sys = np.stack([plant.matrices[j][0, 0, :] for j in range(2)], axis=1)
    srcs = [s.array for s in plan.sources]
    srcs[g["ids"][0]], srcs[g["ids"][1]] = A_sys, B_sys
    given = rng.standard_normal([form.given_len, 1])
    res = plan_emulator.run_resident(plan, given, sources=srcs)
    A, h, Q, q = orc.assemble(form, given)
    for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
        assert_close(res[key], ref, 1e-12, key)


def test_body_case_plan(cpu_api):
    form = problems.body_case(cpu_api)
    rng = np.random.default_rng(3)
    plan = check_plan(form, rng.standard_normal([form.given_len, 1]))
    # the crossed cost expands into two gterms, the others into one per axis
    assert plan.n_gterms == 2 + 1 + 2 + 2
    # index maps are carried bit-exact
    assert plan.optim_ID == {v: form.optim_ID[v] for v in form.optim_variables}


@pytest.mark.parametrize("step_samples", [8, 12])
def test_biped_plan_over_ticks(cpu_api, step_samples):
    conf = problems.BipedConfig(step_samples=step_samples)
    form = problems.biped(cpu_api, conf)
    clock = problems.StepClock(conf.step_samples, form.domain["Ds_x"])
    rng = np.random.default_rng(11)
    widths = set()
    for tick in range(10):
        form.update(step_times=clock.step_times, step_count=clock.step_count)
        given = form.arrange_given(problems.biped_given_collector(form, rng, 0.01))
        if tick in (0, 1, 7, 8, 9):
            it = check_plan(form, given).itab
            if step_samples == 8:
                # one previewed step: x ends and y begins inside one 16-byte piece of every row
                # with two axes -- those pieces are on the list the rounds leave out, every round
                # of the descriptor path stays a one-axis round
                assert (it[_H["RS_NGFIX"]] > 0) == (form.optim_len == 34)
                assert it[_H["RS_GSINGLE"]] == (1 << 24) - 1
        widths.add(form.optim_len)
        clock.tick()
    assert len(widths) == 2      # ragged QP width across walking phases


def test_lipm3d_and_lti_plans(cpu_api):
    form = problems.lipm3d(cpu_api, N=12)
    check_plan(form, np.random.default_rng(0).normal(0, 0.05, [form.given_len, 1]))
    form = problems.random_lti(cpu_api, np.random.default_rng(1), nx=5, nu=2, N=6)
    check_plan(form, np.random.default_rng(2).standard_normal([form.given_len, 1]))


def test_compact_workspace_plans(cpu_api, monkeypatch):
    """Wide problems of several axes keep only each row-set's own columns in the persistent kernel's
    workspace (plan.py Workspace.windows): C3 (three axes, 96 unknowns) with its horizon matrices
    generated on chip shrinks to a third; the emulated program
    (operands by address, windows of the rows of G, descriptors where the problem has them) gives the
    oracle's QP, and the very same plans compiled dense (MPCASM_NO_COMPACT) do too."""
    form = problems.lipm3d(cpu_api, N=32)
    given = np.random.default_rng(0).normal(0, 0.05, [form.given_len, 1])
    A, h, Q, q = orc.assemble(form, given)
    lip = form.dynamics["LIP"]
    sizes = {}
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("MPCASM_NO_COMPACT", "1")
        plan = compile_plan(form, lti=["LIP"])
        it, ws = plan.itab, plan.workspace
        assert it[_H["RS_OK"]] == 1 and it[_H["RS_COMPACT"]] == (0 if dense else 1)
        sizes[dense] = ws.doubles
        if not dense:
            assert (ws.ldv, ws.vd) == (34, 32) and set(ws.c0) == {0, 32, 64}
        g = plan.lti[0]
        srcs = [s.array for s in plan.sources]
        srcs[g["ids"][0]] = lip.matrices[-1][0].T.copy()
        srcs[g["ids"][1]] = lip.matrices[0][0, 0, :].reshape(3, 1).copy()
        res = plan_emulator.run_resident(plan, given, sources=srcs)
        for key, ref in (("P", Q), ("q", q.ravel()), ("G", A), ("h", h.ravel())):
            assert_close(res[key], ref, 1e-12, key)
    assert 2.8 * sizes[False] < sizes[True]
    monkeypatch.delenv("MPCASM_NO_COMPACT")
    # the biped's workspace is small: dense unless asked for (tick by tick, both widths, both
    # layouts: check_plan in test_biped_plan_over_ticks)
    for samples in (12, 8):
        biped = problems.biped(cpu_api, problems.BipedConfig(step_samples=samples))
        biped.update(step_times=np.array([samples - 2, 2 * samples - 2]), step_count=0)
        assert compile_plan(biped).itab[_H["RS_COMPACT"]] == 0
        assert compile_plan(biped, workspace="compact").itab[_H["RS_COMPACT"]] == 1
    # a CSC plan reads the workspace entry by entry: always dense
    biped = problems.biped(cpu_api, problems.BipedConfig(step_samples=12))
    biped.update(step_times=np.array([10, 22]), step_count=0)
    assert compile_plan(biped, csc="upper", workspace="compact").itab[_H["RS_COMPACT"]] == 0
    with pytest.raises(ValueError):
        compile_plan(biped, workspace="tiny")


def test_workspace_chosen_by_what_fits_two_workgroups(cpu_api):
    """``engine.plan_for_device`` (no device needed: ``mpcasm_resident_lds_bytes`` lays the tables out
    as the kernel does): the biped at N = 24 with S, U read from memory needs 95 KB of LDS with the
    dense workspace -- one workgroup per CU -- and 74 KB with the compact one: compact; with its
    horizon matrices built on chip the dense one fits two already: dense; C2 dense; C3 compact."""
    from mpcasm import engine

    def biped(samples):
        form = problems.biped(cpu_api, problems.BipedConfig(step_samples=samples))
        form.update(step_times=np.array([samples - 2, 2 * samples - 2]), step_count=0)
        return form

    n24, c2, c3 = biped(12), biped(8), problems.lipm3d(cpu_api, N=32)
    for form, lti, compact in ((n24, (), 1), (n24, ("LIP",), 0), (c2, (), 0), (c2, ("LIP",), 0),
                               (c3, ("LIP",), 1)):
        plan = engine.plan_for_device(form, lti=lti)
        assert plan.workspace.compact == compact
        direct, in_lds = engine.resident_lds_bytes(plan)
        assert in_lds - direct == 8 * plan.no * (plan.no + plan.no % 2)      # P beside the workspace
        assert direct <= engine.HALF_CU_LDS
        if compact:
            dense = engine.resident_lds_bytes(compile_plan(form, lti=lti, workspace="dense"))
            saved = 8 * (compile_plan(form, lti=lti, workspace="dense").workspace.doubles
                         - plan.workspace.doubles)
            # (what the workspace shrinks by, less the windows of the rows of G: 4 bytes a row)
            assert dense[0] > engine.HALF_CU_LDS and dense[0] - direct == saved - 4 * plan.nc
        # asked for by name, the layout is taken as it is
        assert engine.plan_for_device(form, lti=lti, workspace="dense").workspace.compact == 0
    assert engine.resident_lds_bytes(compile_plan(c3)) == (0, 0)            # not on the persistent kernel
    # an assembler for streaming launches (P collected in LDS) is judged with P in LDS first: the biped
    # at N = 24 with its matrices on chip needs 85 KB dense, 65 KB compact
    assert engine.plan_for_device(n24, lti=("LIP",), batch=16384).workspace.compact == 1
    assert engine.plan_for_device(n24, lti=("LIP",), batch=4096).workspace.compact == 0
    assert engine.plan_for_device(n24, batch=16384).workspace.compact == 1   # (74 KB with P direct)
    assert engine.plan_for_device(c2, lti=("LIP",), batch=65536).workspace.compact == 0


def test_constraint_with_L_and_per_row_fields(cpu_api):
    """Rows from L, per-row arrow / center / extreme, 1-D and 2-D definition
    coefficients, a state-space box."""
    api = cpu_api
    rng = np.random.default_rng(8)
    N = 6
    A, B = problems.random_lti_matrices(rng, 3, 2)
    ext = api.ExtendedSystem.from_cotrol_system(
        api.ControlSystem(["u0", "u1"], ["p", "v", "a"], A, B, axes=["_x", "_y"]), "x", N)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    form.incorporate_definitions({
        "mix_x": api.LineCombo({"p_x": rng.standard_normal((4, N)), "v_x": rng.standard_normal((4, N))}),
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
    check_plan(form, rng.standard_normal([form.given_len, 1]))


def test_param_refresh_and_fingerprint(cpu_api):
    form = problems.body_case(cpu_api)
    plan = compile_plan(form)
    limits = orc.all_limits(form)
    fp = structure_fingerprint(form.goals, limits)
    assert plan.fingerprint == fp
    form.goals["velocity"].update(aim=[3.0, -1.0], weight=7.0)
    limits[0].update(extreme=2.5)
    assert structure_fingerprint(form.goals, limits) == fp      # numbers only
    params = plan.current_params()
    s, r, c = plan.param_slots[("cost", "velocity", "aim")]
    assert list(params[s:s + r * c]) == [3.0, -1.0]
    given = np.random.default_rng(0).standard_normal([form.given_len, 1])
    out = plan_emulator.run(plan, given, params=params)
    A, h, Q, q = orc.assemble(form, given)
    assert_close(out["q"], q.ravel(), 1e-12)
    assert_close(out["h"], h.ravel(), 1e-12)
    form.goals["terminal"].update(schedule=range(7, 9))
    assert structure_fingerprint(form.goals, limits) != fp      # structure changed


def test_invalid_problems_raise(cpu_api):
    api = cpu_api
    form = problems.body_case(api)
    form.goals["velocity"].aim = np.zeros([2, 2])                # two rows: the reference cannot
    with pytest.raises(ValueError):                              # broadcast them either
        compile_plan(form)
    form = problems.body_case(api)
    form.definitions["DCM_x"].variables[0] = "unknown"
    with pytest.raises(KeyError):
        compile_plan(form)


def test_header_slots_agree_with_the_kernels_enum():
    """The table header is written by mpcasm/plan.py and read by csrc/plan_tables.h: same
    slots in the same order, same constants."""
    import os
    import re

    import mpcasm.plan as P

    text = open(os.path.join(os.path.dirname(__file__), "..", "mpc-interface_amd", "csrc",
                             "plan_tables.h")).read()
    body = text[text.index("H_MAGIC = 0"):text.index("H_WORDS =")]
    names = re.findall(r"^\s*H_([A-Z0-9_]+)\s*(?:=\s*\d+\s*)?,", body, flags=re.M)
    assert names == [n for n, _ in sorted(P._H.items(), key=lambda kv: kv[1])]
    for const in ("PLAN_VERSION", "H_WORDS", "RS_TRIP_WORDS", "RS_RR_WORDS", "RS_LTI_WORDS",
                  "RS_BLOCKS_MAX"):
        m = re.search(r"\b%s\s*=\s*(\d+)" % const, text)
        assert m and int(m.group(1)) == getattr(P, const), const


def test_plan_tables_rejected_or_accepted_by_the_library(cpu_api):
    """mpcasm_plan_create validates the tables on the host before touching the
    device: a corrupted plan is MPCASM_ERR_PLAN everywhere, a good one reaches the
    device step (MPCASM_ERR_NODEVICE on this CPU-only machine, OK on a GPU box)."""
    lib = capi.load()
    plan = compile_plan(problems.body_case(cpu_api))
    handle = ctypes.c_void_p()

    def create(itab, dtab):
        return lib.mpcasm_plan_create(itab.ctypes.data, itab.size, dtab.ctypes.data, dtab.size,
                                      ctypes.byref(handle))

    rc = create(plan.itab, plan.dtab)
    assert rc in (0, -4), rc
    if rc == 0:
        assert lib.mpcasm_plan_destroy(handle) == 0
    for word, value in (("MAGIC", 0), ("VERSION", 99), ("OFF_ROWPTR", 10 ** 6), ("NC", -1),
                        ("LDV", 3)):
        bad = plan.itab.copy()
        bad[_H[word]] = value
        assert create(bad, plan.dtab) == -2, word
    bad = plan.itab.copy()
    bad[bad[_H["OFF_ENTBASE"]]] = 10 ** 6          # base id out of range
    assert create(bad, plan.dtab) == -2
    assert create(plan.itab[:-1].copy(), plan.dtab) == -2


def test_persistent_kernel_tables_are_validated(cpu_api):
    """The tables only the persistent kernel reads -- full and short trips, packed row
    records, piece descriptors of G, per-column diagonal terms, the element program of the
    preview matrices -- are checked on the host like the rest: one corrupted word each."""
    import mpcasm.plan as P

    lib = capi.load()
    form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    plan = compile_plan(form)
    handle = ctypes.c_void_p()

    def create(itab):
        return lib.mpcasm_plan_create(itab.ctypes.data, itab.size, plan.dtab.ctypes.data,
                                      plan.dtab.size, ctypes.byref(handle))

    rc = create(plan.itab)
    assert rc in (0, -4), rc
    if rc == 0:
        assert lib.mpcasm_plan_destroy(handle) == 0
    it = plan.itab
    assert it[_H["RS_OK"]] == 1 and it[_H["RS_NGDESC"]] > 0 and it[_H["PM_NFD"]] > 0
    trip0 = it[_H["OFF_RS_TRIP"]]
    shorts = [t for t in range(it[_H["RS_NTRIP"]])
              if (it[trip0 + t * P.RS_TRIP_WORDS + P.RT_WORD] >> P.RT_SHORT) & 1]
    assert shorts                                       # the one-row terminal terms: short trips
    short = trip0 + shorts[0] * P.RS_TRIP_WORDS
    corruptions = {
        "trip: 17 rows": (trip0 + P.RT_WORD, (it[trip0 + P.RT_WORD] & ~31) | 17),
        "trip: A offset off the workspace": (trip0 + P.RT_A, 8 * it[_H["RTOT"]] * it[_H["LDV"]]),
        "trip: weight slot": (trip0 + P.RT_W, 8 * int(it[_H["NPARAMS"]])),
        "trip: block column": (trip0 + P.RT_BJ, 0x7F7F7F7F),
        "short trip: 5 rows": (short + P.RT_WORD, (it[short + P.RT_WORD] & ~31) | 5),
        "short trip: misaligned offset": (short + P.RT_A, it[short + P.RT_A] + 4),
        "trip: d offset not on column no": (trip0 + P.RT_D, it[trip0 + P.RT_D] + 8),
        "wave list": (it[_H["OFF_RS_WTRIP"]] + 1, it[it[_H["OFF_RS_WTRIP"]] + 1] + 2),
        "row record: packed words": (it[_H["OFF_RS_RR"]] + P.RS_RR_WORDS - 2, 0x12340000),
        "piece descriptor of G": (it[_H["OFF_RS_GDESC"]] + 2, it[it[_H["OFF_RS_GDESC"]] + 2] + 2),
        "diagonal term slot": (it[_H["OFF_RS_DPAR"]], int(it[_H["NPARAMS"]]) + 1),
        "preview map": (it[_H["OFF_PM_MAP"]], int(it[_H["PM_NFD"]])),
        "preview op source": (it[_H["OFF_PM_OP"]] + 1, 200),
        "preview op list ends": (it[_H["OFF_PM_FDPTR"]] + int(it[_H["PM_NFD"]]), int(it[_H["PM_NOPS"]]) + 1),
    }
    for what, (word, value) in corruptions.items():
        bad = it.copy()
        assert bad[word] != value, what
        bad[word] = value
        assert create(bad) == -2, what


def test_the_persistent_kernel_compiles_for_a_plan_without_a_device(cpu_api):
    """mpcasm_jit_check: plan tables -> generated constants -> hiprtc (gfx950).  No GPU is
    needed to compile, so the run-time specialisation of the persistent kernel (csrc/jit.hip)
    is build-checked here for the biped (K1 fused and not), the reference's test_body problem
    and a crossed-cost plan."""
    lib = capi.load()
    forms = []
    form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    forms += [(form, ("LIP",)), (form, ())]
    forms.append((problems.body_case(cpu_api), ()))
    log = ctypes.create_string_buffer(1 << 16)
    for form, lti in forms:
        plan = compile_plan(form, lti=lti)
        assert plan.itab[_H["RS_OK"]] == 1
        rc = lib.mpcasm_jit_check(plan.itab.ctypes.data, plan.itab.size, plan.dtab.ctypes.data,
                                  plan.dtab.size, log, len(log))
        assert rc == 0, log.value.decode()
    # a plan the persistent kernel does not take is reported as such, not compiled
    big = compile_plan(problems.random_lti(cpu_api, np.random.default_rng(3), N=16))
    if big.itab[_H["RS_OK"]] == 0:
        assert lib.mpcasm_jit_check(big.itab.ctypes.data, big.itab.size, big.dtab.ctypes.data,
                                    big.dtab.size, None, 0) == -5


def test_compiled_kernels_are_kept_on_disk(cpu_api, tmp_path, monkeypatch):
    """The per-plan code object is named by a hash of everything the compiler sees and kept in
    MPCASM_CACHE_DIR: the second request of the same plan (a later process, another rank) reads
    the file instead of compiling; MPCASM_NO_DISK_CACHE=1 compiles every time."""
    lib = capi.load()
    form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    plan = compile_plan(form, lti=["LIP"])
    monkeypatch.setenv("MPCASM_CACHE_DIR", str(tmp_path / "cache"))

    def stats():
        out = (ctypes.c_int64 * 3)()
        assert lib.mpcasm_jit_stats(out) == 0
        return list(out)

    def check():
        log = ctypes.create_string_buffer(4096)
        return lib.mpcasm_jit_check(plan.itab.ctypes.data, plan.itab.size, plan.dtab.ctypes.data,
                                    plan.dtab.size, log, len(log))

    s0 = stats()
    rc = check()
    if rc == -5:
        pytest.skip("no libhiprtc.so")
    assert rc == 0
    s1 = stats()
    built = s1[0] - s0[0]
    assert built >= 1 and s1[2] - s0[2] == built and s1[1] == s0[1]
    files = sorted((tmp_path / "cache").glob("*.co"))
    assert len(files) == built and all(f.stat().st_size > 1000 for f in files)
    assert check() == 0                                      # ... the second time: from the files
    s2 = stats()
    assert s2[0] == s1[0] and s2[1] - s1[1] == built
    monkeypatch.setenv("MPCASM_NO_DISK_CACHE", "1")
    assert check() == 0
    s3 = stats()
    assert s3[0] - s2[0] == built and s3[1] == s2[1]


@pytest.mark.parametrize("nx,nu,N,kw", [
    (5, 3, 48, {}),
    (12, 6, 64, {}),                                       # the C4 shape: K = 12 terms, 6 column blocks
    (4, 6, 40, dict(scaled=True)),
    (3, 4, 40, dict(extra_unknown=True, scaled=True)),
    (4, 5, 32, dict(given_input=True, two_axis_limit=True)),
])
def test_scan_form_tables_against_the_oracle(cpu_api, nx, nu, N, kw):
    """The tiled kernel's scan form (P summed along diagonals, plan_tables.h T_SCAN*), emulated from
    the plan's tables for a system of its own per instance, against the oracle on the reference's
    dense matrices of that system."""
    rng = np.random.default_rng(nx * 100 + N)
    form, _, _ = lti_tracking_problem(cpu_api, rng, nx, nu, N, **kw) if kw else (
        problems.random_lti(cpu_api, rng, nx=nx, nu=nu, N=N), None, None)
    plan = compile_plan(form, lti=["plant"])
    it = plan.itab
    assert it[_H["T_OK"]] == 1 and it[_H["T_TOEPLITZ"]] == 1 and it[_H["T_SCAN"]] == nx
    assert it[_H["T_SCAN_NBLK"]] == nu - (1 if kw.get("given_input") else 0)
    assert it[_H["T_SCAN_NOTHER"]] == (N if kw.get("extra_unknown") else 0)
    assert it[_H["T_SCAN_NGREST"]] == (N if kw.get("two_axis_limit") else 0)
    A, B = problems.random_lti_matrices(rng, nx, nu)
    given = rng.normal(0, 0.3, form.given_len)
    out = plan_emulator.run_scan(plan, given, ab=[(A, B)])
    # the set-up fused into the kernel (H_T_SCAN_FUSED): `given` is the initial state and nothing else, every
    # workspace row a row of one of the terms, no row of G through the column tables -- then
    # d[first row of the term + k] = c (A^{k+1} x0)[state], which is what the kernel writes instead of reading d
    fused = int(it[_H["T_SCAN_FUSED"]])
    assert fused == (1 if not kw else 0)      # (the others: a given input; a limit on rows no cost has; a row of G over two rows)
    if fused:
        K = int(it[_H["T_SCAN"]])
        gt = it[it[_H["OFF_T_SCAN_GT"]]:it[_H["OFF_T_SCAN_GT"]] + 4 * K].reshape(K, 4)
        gc = plan.dtab[it[_H["T_DOFF_SCAN_GC"]]:it[_H["T_DOFF_SCAN_GC"]] + K]
        d, x = np.zeros(K * N), np.array(given, dtype=float)
        for k in range(N):
            x = A @ x
            for g in range(K):
                d[gt[g, 3] + k] = gc[g] * x[gt[g, 0] // (nu * N)]
        assert_close(d, out["ref"]["d"][:K * N], 1e-13, "d from the free response")
        assert plan.itab[_H["RTOT"]] == K * N
    dyn = form.dynamics["plant"]
    saved = list(dyn.matrices)
    try:
        So, Uo = orc.extend_matrices(N, A, B)
        dyn.matrices = Uo + [So]
        dyn.update_definitions()
        Ao, ho, Qo, qo = orc.assemble(form, given.reshape(-1, 1))
    finally:
        dyn.matrices = saved
        dyn.update_definitions()
    for key, ref in (("P", Qo), ("q", qo.ravel()), ("G", Ao), ("h", ho.ravel())):
        assert_close(out[key], ref, 1e-12, "scan " + key)


def test_scan_form_only_where_it_holds(cpu_api):
    """No scan form: a cost on part of the horizon (the diagonal sums would need the missing rows), a
    horizon beyond the kernel's lanes, sources read from memory (no generated group)."""
    rng = np.random.default_rng(3)
    form, _, _ = lti_tracking_problem(cpu_api, rng, 4, 4, 40, scheduled_cost=True)
    plan = compile_plan(form, lti=["plant"])
    assert plan.itab[_H["T_TOEPLITZ"]] == 1 and plan.itab[_H["T_SCAN"]] == 0
    form = problems.random_lti(cpu_api, rng, nx=3, nu=2, N=100)
    assert compile_plan(form, lti=["plant"]).itab[_H["T_SCAN"]] == 0
    form = problems.random_lti(cpu_api, rng, nx=4, nu=4, N=40)
    assert compile_plan(form).itab[_H["T_SCAN"]] == 0


def test_scan_form_tables_are_validated(cpu_api):
    """One corrupted word of the scan tables each: MPCASM_ERR_PLAN before anything reaches a device."""
    import mpcasm.plan as P

    lib = capi.load()
    rng = np.random.default_rng(4)
    form, _, _ = lti_tracking_problem(cpu_api, rng, 4, 5, 32, given_input=True, two_axis_limit=True)
    plan = compile_plan(form, lti=["plant"])
    it = plan.itab
    handle = ctypes.c_void_p()

    def create(itab):
        return lib.mpcasm_plan_create(itab.ctypes.data, itab.size, plan.dtab.ctypes.data,
                                      plan.dtab.size, ctypes.byref(handle))

    rc = create(it)
    assert rc in (0, -4), rc
    if rc == 0:
        assert lib.mpcasm_plan_destroy(handle) == 0
    nparams = int(it[_H["NPARAMS"]])
    blk, gt, grow = it[_H["OFF_T_SCAN_BLK"]], it[_H["OFF_T_SCAN_GT"]], it[_H["OFF_T_SCAN_GROW"]]
    simple = next(R for R in range(plan.nc) if it[grow + 2 * R] >= 0)
    corruptions = {
        "terms": (_H["T_SCAN"], P.T_SCAN_KMAX + 1),
        "blocks": (_H["T_SCAN_NBLK"], 0),
        "block: first column": (blk, plan.no),
        "block: input offset": (blk + 1, it[blk + 1] + 1),
        "term: state offset": (gt, it[gt] + 1),
        "term: weight slot": (gt + 1, nparams),
        "term: rows of d": (gt + 3, plan.rtot),
        "row of G: table row": (grow + 2 * simple, it[grow + 2 * simple] + 32),
        "row of G: arrow": (grow + 2 * simple + 1, it[grow + 2 * simple + 1] + 1),
        "rows of G by the tables: count": (_H["T_SCAN_NGREST"], it[_H["T_SCAN_NGREST"]] - 1),
        "unknowns outside the blocks": (_H["T_SCAN_NOTHER"], 1),
        "column -> block": (it[_H["OFF_T_SCAN_COLBLK"]], 1),
    }
    for what, (word, value) in corruptions.items():
        bad = it.copy()
        assert bad[word] != value, what
        bad[word] = value
        assert create(bad) == -2, what


def test_causality_of_bound_horizon_matrices_is_part_of_the_contract(cpu_api):
    """The tile masks of the tiled kernel (and the CSC patterns) skip U_j[k][l > k]: a plan records the
    sources it took as causal (``Plan.causal_assumed``), and a formulation whose U_j holds entries above
    the diagonal compiles to a plan that assumes nothing of them."""
    from mpcasm.plan import is_causal

    rng = np.random.default_rng(8)
    form = problems.random_lti(cpu_api, rng, nx=4, nu=4, N=32)            # no = 128: the tiled kernel
    plan = compile_plan(form)
    dyn = form.dynamics["plant"]
    u_ids = [i for i, s in enumerate(plan.sources) if s.key[0] == "plant" and s.key[1] < 4]
    assert plan.itab[_H["T_OK"]] == 1 and plan.causal_assumed == u_ids
    assert all(is_causal(plan.sources[i].array) for i in u_ids)
    assert compile_plan(form, lti=["plant"]).causal_assumed == []         # generated on chip: causal by construction
    small = problems.random_lti(cpu_api, rng, nx=3, nu=2, N=8)            # the persistent kernel: structural masks only
    assert compile_plan(small).causal_assumed == []
    # entries above the diagonal: every tile of that source counts as non-zero
    saved = [np.array(M) for M in dyn.matrices]
    try:
        dyn.matrices[1][3, 20, 0] = 0.25
        dyn.update_definitions()
        form.make_preview_matrices()
        dense = compile_plan(form)
        assert u_ids[1] not in dense.causal_assumed and not is_causal(dense.sources[u_ids[1]].array)
        given = rng.normal(0, 0.3, [form.given_len, 1])
        out = plan_emulator.run_tiled(dense, given)                       # (asserts: nothing outside the multiplied tiles)
        A, h, Q, q = orc.assemble(form, given)
        assert_close(out["P"], Q, 1e-12, "P"), assert_close(out["G"], A, 1e-12, "G")
    finally:
        for M, old in zip(dyn.matrices, saved):
            M[...] = old
        dyn.update_definitions()
        form.make_preview_matrices()


@pytest.mark.parametrize("N", [12, 100])
def test_sweep_tables_against_the_oracle(cpu_api, N):
    """A dynamics compiled as ``ltv`` (per-step A_k, B_k; BASELINE config C5): the sweep kernel's
    recursions, emulated from the plan's tables, against the oracle on the dense S, U of
    ``extend_matrices_ltv`` for the same per-step system -- and, pinned to the reference, with every
    step the formulation's own pair against the oracle on the reference's ``extend_matrices``."""
    form = problems.lipm_ltv(cpu_api, N=N)
    plan = compile_plan(form, ltv=["LIP"])
    assert plan.itab[_H["SW_OK"]] == 1 and plan.no == 2 * N and plan.nc == 4 * N + 4
    rng = np.random.default_rng(N)
    given = rng.normal(0, 0.05, form.given_len)
    dyn = form.dynamics["LIP"]
    saved = list(dyn.matrices)
    A, B = problems.ltv_lipm_steps(cpu_api, N=N, theta=0.4)
    A0 = np.broadcast_to(saved[-1][0].T, A.shape)
    B0 = np.broadcast_to(saved[0][0, 0, :].reshape(-1, 1), B.shape)
    try:
        for Ak, Bk, extend in ((A, B, lambda: orc.extend_matrices_ltv(N, A, B)),
                               (A0, B0, lambda: orc.extend_matrices(N, A0[0], B0[0]))):
            out = plan_emulator.run_sweep(plan, given, Ak, Bk)
            So, Uo = extend()
            dyn.matrices = list(Uo) + [So]
            dyn.update_definitions()
            Ao, ho, Qo, qo = orc.assemble(form, given.reshape(-1, 1))
            for key, ref in (("P", Qo), ("q", qo.ravel()), ("G", Ao), ("h", ho.ravel())):
                assert_close(out[key], ref, 1e-12, "sweep " + key)
    finally:
        dyn.matrices = saved
        dyn.update_definitions()


def test_what_cannot_be_assembled_step_by_step_is_refused(cpu_api):
    """``ltv=`` needs every row of a cost or limit to be one step's states: a crossed cost, an L that
    mixes steps, an unknown that is no input of the system -- ValueError at compile time."""
    api = cpu_api
    N = 12
    form = problems.lipm_ltv(api, N=N)
    form.incorporate_goal("crossed", api.Cost("CoM", 0.1, aim=[0, 0], axes=["_x", "_y"], cross="CoM_dot",
                                              cross_aim=[0, 0]))
    with pytest.raises(ValueError, match="crossed"):
        compile_plan(form, ltv=["LIP"])
    form = problems.lipm_ltv(api, N=N)
    form.incorporate_goal("mixed", api.Cost("CoM", 0.1, aim=[0, 0], axes=["_x", "_y"],
                                            L=[np.tril(np.ones((N, N)))] * 2))
    with pytest.raises(ValueError, match="one step"):
        compile_plan(form, ltv=["LIP"])
    form = problems.biped(api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    with pytest.raises(ValueError):
        compile_plan(form, ltv=["LIP"])


def test_sweep_tables_are_validated(cpu_api):
    import mpcasm.plan as P

    lib = capi.load()
    form = problems.lipm_ltv(cpu_api, N=12)
    plan = compile_plan(form, ltv=["LIP"])
    it = plan.itab
    handle = ctypes.c_void_p()

    def create(itab):
        return lib.mpcasm_plan_create(itab.ctypes.data, itab.size, plan.dtab.ctypes.data,
                                      plan.dtab.size, ctypes.byref(handle))

    rc = create(it)
    assert rc in (0, -4), rc
    if rc == 0:
        assert lib.mpcasm_plan_destroy(handle) == 0
    ax, tr, lm, col = (it[_H[k]] for k in ("OFF_SW_AXIS", "OFF_SW_TERM", "OFF_SW_LIM", "OFF_SW_COL"))
    corruptions = {
        "states": (_H["SW_N"], P.SW_NMAX + 1),
        "source slot": (_H["SW_SRC_A"], int(it[_H["NSRC"]])),
        "axis: initial state": (ax, plan.ng),
        "axis: first unknown": (ax + 1, plan.no - 3),
        "column word": (col + 5, it[col + 5] + (1 << 16)),
        "term: step": (tr + 1, 12),
        "term: weight": (tr + 4, int(it[_H["NPARAMS"]])),
        "term: combination": (tr + 6, 4 * int(it[_H["SW_NCVEC"]])),
        "limit: first line": (lm, plan.nc),
        "limit: axis": (lm + P.SW_LIM_WORDS, 4),
        "limit: arrow": (lm + P.SW_LIM_WORDS + 4, int(it[_H["NPARAMS"]])),
        "limit: steps of its axes": (lm + P.SW_LIM_WORDS + P.SW_LAX_WORDS + 1, 3),
    }
    for what, (word, value) in corruptions.items():
        bad = it.copy()
        assert bad[word] != value, what
        bad[word] = value
        assert create(bad) == -2, what


def test_an_odd_width_compiles_for_the_tiled_kernel(cpu_api):
    """129 unknowns: the plan is one the tiled kernel takes (T_OK; round 3 required an even width because rows
    of G leave in 16-byte pieces -- the general form now writes the rows of an odd width with 8-byte stores),
    the library accepts its tables, and the tables give the oracle's numbers."""
    from mpcasm import engine

    rng = np.random.default_rng(5)
    form = problems.random_lti(cpu_api, rng, nx=3, nu=3, N=43)
    plan = compile_plan(form)
    assert plan.no == 129 and plan.itab[_H["T_OK"]] == 1 and plan.itab[_H["T_NOP"]] == 256
    assert engine.resident_lds_bytes(plan) == (0, 0)          # validated like mpcasm_plan_create does; not persistent
    given = rng.normal(0, 0.3, [form.given_len, 1])
    out = plan_emulator.run_tiled(plan, given)
    A, h, Q, q = orc.assemble(form, given)
    assert_close(out["P"], Q, 1e-12, "P"), assert_close(out["q"], q.ravel(), 1e-12, "q")
    assert_close(out["G"], A, 1e-12, "G"), assert_close(out["h"], h.ravel(), 1e-12, "h")


@pytest.mark.parametrize("kw", [{}, dict(scaled=True, two_axis_limit=True), dict(extra_unknown=True, scheduled_cost=True),
                                dict(given_input=True)], ids=["plain", "scaled+two-axis", "slack+schedule", "given input"])
def test_what_the_shared_model_form_relies_on(cpu_api, kw):
    """The shared-model form of the tiled kernel (csrc/tiled.hip launch_shared_form) writes P_b = sum_g w_b[g] K_g
    with K_g the Hessian at "weight slot g = 1, every other parameter 0", and takes aims, arrows, centers and
    extremes for irrelevant to P.  On the plan's own tables (the emulator of the general form): P is exactly that
    sum over the weight slots of the stages and of the diagonal terms -- with every parameter random -- and G is
    linear in the arrows alone."""
    from helpers import lti_tracking_problem
    from mpcasm.plan import GT_WORDS

    rng = np.random.default_rng(31)
    form, _, _ = lti_tracking_problem(cpu_api, rng, 4, 5 if kw.get("given_input") else 4, 32, **kw)
    plan = compile_plan(form)
    it = plan.itab
    assert it[_H["T_OK"]] == 1
    given = rng.normal(0, 0.3, [plan.ng])
    params = np.array(plan.params, dtype=float) * rng.uniform(0.5, 1.5, plan.params.shape) + rng.normal(0, 0.1, plan.params.shape)
    full = plan_emulator.run_tiled(plan, given, params=params)
    # the weight slots as the launcher collects them: every stage's, every diagonal term's
    nstage, off = int(it[_H["T_NSTAGE"]]), int(it[_H["OFF_T_STAGE"]])
    slots = {int(it[off + s * 16 + 4]) for s in range(nstage)}                   # TS_WPARAM of T_STAGE_WORDS = 16
    ngt, goff = int(it[_H["NGTERM"]]), int(it[_H["OFF_GTERM"]])
    slots |= {int(it[goff + g * GT_WORDS + 3]) for g in range(ngt) if it[goff + g * GT_WORDS + 6] & 4}   # GT_WPARAM of DIAG terms
    assert 0 < len(slots) <= 32
    P = np.zeros_like(full["P"])
    for slot in sorted(slots):
        unit = np.zeros_like(params)
        unit[slot] = 1.0
        P += params[slot] * plan_emulator.run_tiled(plan, np.zeros(plan.ng), params=unit)["P"]
    assert_close(P, full["P"], 1e-13, "P as the weighted sum of the per-weight Hessians")
    # G: no weight, aim, center or extreme in it
    others = params.copy()
    arrow_cols = np.zeros(len(params), dtype=bool)
    for (kind, name, field), (start, rows, cols) in plan.param_slots.items():
        if field == "arrow":
            arrow_cols[start:start + rows * cols] = True
    others[~arrow_cols] = rng.normal(0, 1, (~arrow_cols).sum())
    assert_close(plan_emulator.run_tiled(plan, given, params=others)["G"], full["G"], 1e-14, "G depends on the arrows alone")
