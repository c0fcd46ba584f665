"""CPU: the drop-in description API (Cost, Constraint, Box, DomainVariable,
ExtendedSystem, ControlSystem, LineCombo, tools) -- constructor defaults, update
rules and error behaviour the reference's own unit tests pin
(python/tests/test_goal.py, test_restrictions.py, test_dynamics.py), plus the
geometry and named-system golden vectors (G4, G5).
"""
import numpy as np
import pytest

from helpers import assert_close, golden
from mpc_interface.combinations import LineCombo
from mpc_interface.dynamics import ControlSystem, DomainVariable, ExtendedSystem
from mpc_interface.goal import Cost
from mpc_interface.restrictions import Box, Constraint, box_boundaries
import mpc_interface.tools as tools


# ---------------------------------------------------------------- Cost (test_goal.py)
def test_cost_constructor_aliasing():
    c1 = Cost("CoM", 8)
    assert c1.variable == "CoM" and c1.aim == 0 and c1.aim is c1.cross_aim
    assert c1.L == [] and c1.L is c1.cross_L and c1.axes == [""] and c1.t is None
    c2 = Cost("CoM", 8, aim=[8, 0], axes=["_x", "_y"])
    assert (c2.aim == [8, 0]).all() and c2.aim.shape == (1, 2) and c2.aim is c2.cross_aim
    c3 = Cost("f", 4, cross="u")
    assert c3.cross == "u" and c3.crossed and c3.cross_L == [] and c3.aim is not c3.cross_aim
    c4 = Cost("f", 4, cross="u", axes=["_x", "_y"], cross_L=[5, 2])
    assert c4.cross_L == [5, 2] and not c4.L
    with pytest.raises(TypeError):
        Cost("f", 1, axes="_x")


def test_cost_update_rules():
    c1 = Cost("CoM", 8)
    c1.update(aim=9)
    c1.update(weight=0)
    c1.update(L=4 * np.eye(3))
    assert c1.cross_aim == 9 and c1.cross_L is c1.L and c1.weight == 0
    with pytest.raises(KeyError):
        c1.update(cross_aim=3)
    with pytest.raises(KeyError):
        c1.update(cross_L=np.eye(3))
    c3 = Cost("f", 4, cross="u")
    c3.update(cross_aim=2)
    assert c3.cross_aim != c3.aim
    c4 = Cost("f", 4, cross="u", axes=["_x", "_y"], cross_L=[5, 2])
    c4.update(cross_L=[])
    assert not c4.cross_L and not c4.L
    c4.update(L=[3, 6])
    assert c4.L and not c4.cross_L
    with pytest.raises(IndexError):
        Cost("f", 1, axes=["_x", "_y"], L=[np.eye(2)] * 3)
    with pytest.raises(ValueError):
        Cost("f", 1, L=np.eye(3), schedule=range(0, 2))
    assert Cost("f", 1, schedule=range(2, 5)).t == 3


# ------------------------------------------------- Constraint (test_restrictions.py)
def test_constraint_constructor_and_normalize():
    c = Constraint("CoM", 4)
    assert c.axes == [""] and c.arrow.shape == (1, 1) and c.center.shape == (1, 1)
    assert c.nlines is None and c.m is None and c.t is None
    g = golden("g4_geometry")
    c2 = Constraint("CoM", [-2, 3], axes=["_x", "_y"], arrow=[[1, 0], [0, -1]])
    assert_close(c2.arrow, g["flipped/arrow"], 0)          # sign flipped so extreme >= 0
    assert_close(c2.extreme, g["flipped/extreme"], 0)
    c3 = Constraint("CoM", [2, 3], axes=["_x", "_y"], arrow=[[1, 0], [0, -1]], center=[1, 1])
    assert_close(c3.bound(), g["twoaxes/bound"], 0)
    assert c3.nlines == 2
    with pytest.raises(ValueError):
        Constraint("CoM", 1, axes=["_x", "_y"])             # arrow required for several axes
    with pytest.raises(TypeError):
        Constraint("CoM", 1, axes="_x")
    with pytest.raises(ValueError):
        Constraint("CoM", [1, 2, 3], arrow=[[1], [1]])      # 3 extremes against 2 arrows


def test_constraint_update_and_rows():
    c = Constraint("s", 10, arrow=[1, 1], axes=["_x", "_y"], L=np.eye(9))
    assert c.m == 9 and c.nlines == 9 and len(c.L) == 2
    c.update(schedule=range(3, 6), L=np.ones((2, 3)))
    assert c.t == 3 and c.m == 2 and c.nlines == 2
    c.update(L=[])
    assert c.nlines == 3                                    # falls back to the schedule
    c.update(schedule=range(0))
    assert c.nlines is None
    c.update(center=np.arange(8).reshape(4, 2))
    assert c.nlines == 4
    with pytest.raises(ValueError):
        c.update(extreme=[1, 2, 3])
    mats = Constraint("v", 2, axes=["_x", "_y"], arrow=[3, 4], L=np.eye(2)).matrices()
    assert_close(mats[0], 3 * np.eye(2), 0)
    assert_close(mats[1], 4 * np.eye(2), 0)
    pts = np.array([[0.0, 0.0], [5.0, 5.0]])
    ok = Constraint("v", 2, axes=["_x", "_y"], arrow=[1, 0]).is_feasible(pts, "TS")
    assert np.ravel(ok).tolist() == [True, False]


def test_box_geometry_matches_reference():
    """Facet order is Qhull's (SURVEY.md section 8a quirk vii): fixtures from the reference."""
    g = golden("g4_geometry")
    foot = tools.make_simetric_vertices(np.array([0.1, 0.05]))
    cuboid = np.array([[sx * 0.2, sy * 0.1, sz * 0.05]
                       for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)]) + 0.3
    boxes = {
        "foot2d": Box.task_space("v", foot, ["_x", "_y"]),
        "stepping2d": Box.task_space(
            "v", tools.make_simetric_vertices(np.array([0.3, 0.1])), ["_x", "_y"]),
        "diamond2d": Box.task_space("v", np.array([[0, 1], [1, 0], [0, -1], [-1, 0]]),
                                    ["_x", "_y"]),
        "cuboid3d": Box.task_space("v", cuboid, ["_x", "_y", "_z"]),
        "segment1d": Box.task_space("v", np.array([[-0.2], [0.5]])),
        "state_space": Box.state_space(
            "v", np.array([[0.0, 1], [1, 0.5], [0.2, -1], [-1, 0]]), schedule=range(0, 2)),
        "foot2d_margin": Box.task_space("v", foot, ["_x", "_y"]),
    }
    boxes["foot2d_margin"].set_safety_margin(0.02)
    assert len(boxes["foot2d"].constraints) == 4 and len(boxes["cuboid3d"].constraints) == 12
    for name, box in boxes.items():
        assert len(box.constraints) == int(g[name + "/n"])
        for k, limit in enumerate(box.constraints):
            p = "%s/%d/" % (name, k)
            assert_close(limit.arrow, g[p + "arrow"], 1e-15, p + "arrow")
            assert_close(limit.center, g[p + "center"], 1e-15, p + "center")
            assert_close(limit.extreme, g[p + "extreme"], 1e-15, p + "extreme")
            assert_close(limit.bound(), g[p + "bound"], 1e-15, p + "bound")
            if p + "L" in g:
                assert_close(np.stack(limit.L), g[p + "L"], 1e-15)


def test_box_transforms():
    foot = tools.make_simetric_vertices(np.array([0.1, 0.05]))
    box = Box.task_space("v", foot, ["_x", "_y"])
    inside, outside = np.array([[0.05, 0.0]]), np.array([[0.5, 0.0]])
    assert box.is_feasible([inside, outside], "TS") == [True, False]
    box.recenter_in_TS([0.5, 0.0])
    assert box.is_feasible([inside, outside], "TS") == [False, True]
    box.translate_in_TS(np.array([-0.5, 0.0]))
    assert box.is_feasible([inside], "TS") == [True]
    box.scale_box(0.1)
    assert box.is_feasible([inside], "TS") == [False]
    box.scale_box(1.0)
    box.rotate_in_TS(tools.rotation2D(np.pi / 2))
    assert box.is_feasible([np.array([[0.0, 0.09]])], "TS") == [True]
    box.reschedule(range(2, 4))
    assert all(l.schedule == range(2, 4) for l in box.constraints)
    with pytest.raises(ValueError):
        box.recenter_in_SS(np.zeros(3))
    with pytest.raises(NotImplementedError):
        box.rotate_in_SS(None)
    arrows, extremes, center = box_boundaries(foot)
    assert_close(np.linalg.norm(arrows, axis=1), np.ones(4), 1e-15)
    assert_close(center, [0, 0], 1e-15)


# ------------------------------------------------------------ dynamics (test_dynamics.py)
def test_domain_variable():
    d = DomainVariable("n", 20, ["_x", "_y"])
    assert d.domain == {"n_x": 20, "n_y": 20} and d.all_variables is d.domain
    assert (d.definitions["n_x"].matrices[0] == np.eye(20)).all()
    d3 = DomainVariable(["n", "H"], [20, 120], ["_x", "_y"])
    assert list(d3.domain.items()) == [("n_x", 20), ("H_x", 120), ("n_y", 20), ("H_y", 120)]
    assert DomainVariable("o", [20]).domain == {"o": 20}
    assert DomainVariable("o", [20], 5).domain == {"o5": 20}
    with pytest.raises(TypeError):
        DomainVariable(3, 2)
    with pytest.raises(IndexError):
        DomainVariable(["a", "b"], [1])

    def resize(var, **kargs):
        var.domain.update({v: kargs["new_sizes"] for v in var.domain_ID})

    dv = DomainVariable("non_lin", 20, ["_x", "_y"], time_variant=True, how_to_update_size=resize)
    dv.define_output("twice", {"non_lin": 2})
    dv.update(new_sizes=7)
    assert dv.domain["non_lin_x"] == 7 and dv.definitions["non_lin_y"].matrices[0].shape == (7, 7)
    frozen = DomainVariable("n", 20, ["_x"], time_variant=False, how_to_update_size=resize)
    frozen.update(new_sizes=7)
    assert frozen.domain["n_x"] == 20


def test_extended_system_bookkeeping(cpu_api):
    inputs = ["u%d" % i for i in range(6)]
    states = ["s%d" % i for i in range(8)]
    A = np.zeros([8, 8])
    A[-1] = 1
    A[:-1, 1:] = np.eye(7)
    B = np.ones([8, 6])
    axes = ["_x", "_y", "_z", "_a", "_b"]

    def how(cnt, **kargs):
        cnt.B = cnt.A @ cnt.B * kargs["factor"]

    live = ControlSystem(inputs, states, A, B, axes, time_variant=True, how_to_update_matrices=how)
    fixed = ControlSystem(inputs, states, A, B, axes, time_variant=False, how_to_update_matrices=how)
    e1 = ExtendedSystem.from_cotrol_system(live, "x", 20)
    e2 = ExtendedSystem.from_cotrol_system(fixed, "e", 20)
    assert len(e1.state_ID) == 8 * 5 and len(e2.domain_ID) == 6 * 5 + 5
    assert len(e1.all_variables) == len(e1.definitions) == 8 * 5 + 6 * 5 + 5
    assert len(e1.matrices) == 7 and e1.matrices[-1].shape == (20, 8, 8)
    assert e1.matrices[0].shape == (20, 20, 8)
    assert list(e1.domain_ID.items())[:8] == [("u0_x", 0), ("u1_x", 1), ("u2_x", 2), ("u3_x", 3),
                                              ("u4_x", 4), ("u5_x", 5), ("x0_x", 6), ("u0_y", 0)]
    U0, S0 = e1.matrices[0], e1.matrices[-1]
    e1.update(control_system=live, factor=7)
    e2.update(control_system=fixed, factor=7)
    assert (e1.matrices[0] != U0).any() and (e1.matrices[-1] == S0).all()
    assert (e2.matrices[0] == U0).all()
    # states only combine same-axis domain variables (dynamics.py:284-294)
    assert e1.definitions["s3_y"].variables == ["u%d_y" % i for i in range(6)] + ["x0_y"]
    assert np.shares_memory(e1.definitions["s3_y"].matrices[0], e1.matrices[0]) or \
        (e1.definitions["s3_y"].matrices[0] == e1.matrices[0][..., 3]).all()
    e1.define_output("new_var", {"u5": 2, "x0": 1})
    assert len(e1.outputs) == 5 and "new_var_x" in e1.definitions
    single = ExtendedSystem("Ds", "s", "s", S=np.ones([9, 1]), U=np.ones([9, 2]), axes=["_x"])
    assert single.matrices[0].shape == (9, 2, 1) and single.matrices[1].shape == (9, 1, 1)
    assert single.domain == {"Ds_x": 2, "s0_x": 1}
    with pytest.raises(TypeError):
        ExtendedSystem("u", "s", 3, np.ones([2, 1]), np.ones([2, 1]))


def test_named_systems_match_reference():
    """G5: closed-form discretisation against the reference's sympy lambdas."""
    g = golden("g5_systems")
    for name in ("P->CC", "P->X", "dP->CCC", "dP->CCP", "J->CCC"):
        get_A, get_B, params = tools.get_system_matrices(name)
        for tau, omega in ((0.1, 3.5), (0.1, 3.3445), (0.05, 3.0)):
            key = "%s/tau%g_omega%g/" % (name.replace("->", "_to_"), tau, omega)
            A = np.asarray(get_A(tau=tau, omega=omega), dtype=float)
            B = np.asarray(get_B(tau=tau, omega=omega), dtype=float)
            assert A.shape == g[key + "A"].shape and B.shape == g[key + "B"].shape
            assert_close(A, g[key + "A"], 1e-13, key + "A")
            assert_close(B, g[key + "B"], 1e-13, key + "B")
            # generic numeric discretisation agrees with the closed forms
        inputs, states = tools.get_system_variables(name)
        assert len(states) == A.shape[0] and len(inputs) == 1
    lip = ControlSystem.from_name("J->CCC", ["_x", "_y"], tau=0.1, omega=3.5)
    assert lip.parameters == {"tau": 0.1, "omega": 3.5} and lip.system_name == "J->CCC"
    with pytest.raises(ValueError):
        ControlSystem.from_name("P->CC", tau=0.1)             # omega missing
    G = np.array([[0, 1.0], [3.5 ** 2, 0]])
    A, B = tools.discretize(G, np.array([0, -3.5 ** 2]), tau=0.1)
    assert_close(A, g["P_to_CC/tau0.1_omega3.5/A"], 1e-12)
    assert_close(B, g["P_to_CC/tau0.1_omega3.5/B"], 1e-12)


# --------------------------------------------------------------------- tools, LineCombo
def test_step_planning_helpers():
    E = tools.plan_steps(9, 1, step_times=np.array([2, 5, 8]))
    assert E.shape == (9, 3) and E.dtype.kind == "i"
    assert E[:, 0].tolist() == [0, 0, 1, 1, 1, 1, 1, 1, 1]
    assert tools.plan_steps(16, 0, regular_time=8).shape == (16, 2)      # steps at t = 0 and 8
    assert tools.plan_steps(16, 0, step_times=np.array([6, 14])).shape == (16, 2)
    with pytest.raises(KeyError):
        tools.plan_steps(4)
    assert tools.n_predicted_steps(0, 16, np.array([7, 15])) == 1
    c = tools.find_step_centers(0, 3, [0.0, 0.28])
    assert_close(c, [[0, -0.28], [0, 0.28], [0, -0.28]], 0)
    assert_close(tools.find_step_centers(1, 2, [0.0, 0.28]), [[0, 0.28], [0, -0.28]], 0)
    assert_close(tools.make_simetric_vertices([2, 1]), [[2, 1], [-2, 1], [-2, -1], [2, -1]], 0)
    assert_close(tools.rotation3D(0.3, "z")[:2, :2], tools.rotation2D(0.3), 0)
    L = tools.step_average_velocity(0, 8, np.array([3]))
    assert L.shape == (1, 16) and abs(L[0, 3] + 0.25) < 1e-15 and abs(L[0, 7] - 0.25) < 1e-15


def test_line_combo():
    combo = LineCombo({"a": 1, "b": np.eye(2)})
    assert combo.keys() == ["a", "b"] and combo["a"] == 1 and len(list(combo.items())) == 2
    assert repr(combo) == "C0 ( a ) + C1 ( b )"
    hits = []
    live = LineCombo({"a": 1}, time_variant=True, how_to_update=lambda c, **k: hits.append(k))
    live.update(x=3)
    LineCombo({"a": 1}, how_to_update=lambda c, **k: hits.append(k)).update(x=4)
    assert hits == [{"x": 3}]


def test_formulation_bookkeeping(cpu_api):
    """identify_qp_domain / update bookkeeping and the incorporation asserts
    (body.py:38-136) -- host logic, no kernel involved."""
    from mpcasm import problems

    form = problems.body_case(cpu_api)
    assert form.optim_variables == ["x0_x", "x0_y", "Ds_x", "Ds_y", "CoM_dddot_x", "CoM_dddot_y"]
    assert set(form.domain) == set(form.optim_variables + form.given_variables)
    assert form.optim_len == 3 + 3 + 3 + 3 + 9 + 9 and form.given_len == 1 + 1 + 9 + 9
    assert form.optim_ID["Ds_x"] == range(6, 9) and form.given_ID["n_y"] == range(11, 20)
    assert form.domain_ID["optim_ID"] is form.optim_ID
    assert len(form.PM) == len(form.definitions)
    with pytest.raises(AssertionError):
        form.incorporate_goal("bad", cpu_api.Cost("nope", 1))
    with pytest.raises(AssertionError):
        form.incorporate_constraint("bad", cpu_api.Constraint("nope", 1))
    given = form.arrange_given({v: np.full([len(r), 1], float(i))
                                for i, (v, r) in enumerate(form.given_ID.items())})
    assert given.shape == (form.given_len, 1) and given[form.given_ID["n_x"]].max() == 2.0
    # without a GPU the numeric path refuses to run: there is no CPU fallback
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            form.generate_all_qp_matrices(given)
        with pytest.raises(RuntimeError):
            form.PM["CoM_x"]


def test_compiled_plans_are_kept_least_recently_used(cpu_api, monkeypatch):
    """Formulation._assembler keeps compiled plans by structure: room for the whole problem and
    every single cost / limit in three structures (what a per-part sweep over the walking loop's
    phases needs), the least recently used entry goes first."""
    import mpcasm.engine as engine
    from mpcasm import problems

    built = []

    class FakeAssembler:
        def __init__(self, form, batch=1, device=None, costs=None, limits=None):
            built.append((None if costs is None else tuple(costs), None if limits is None else len(limits)))

        def rebind_sources(self, form, frozen):
            return True

        def refresh_params(self):
            return True

    monkeypatch.setattr(engine, "Assembler", FakeAssembler)
    form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    parts = 1 + len(form.goals) + len(form._all_limits())
    assert form._asm_cache_size() == max(24, 3 * parts) and parts == 19
    form._ASM_CACHE_MIN = 3                                     # a small cache for the test
    monkeypatch.setattr(type(form), "_asm_cache_size", lambda self: 3)
    names = list(form.goals)
    a = form._assembler(costs={"cost": form.goals[names[0]]})
    b = form._assembler(costs={"cost": form.goals[names[1]]})
    c = form._assembler(costs={"cost": form.goals[names[2]]})
    assert len(built) == 3 and len(form._asm_cache) == 3
    assert form._assembler(costs={"cost": form.goals[names[0]]}) is a and len(built) == 3   # a hit
    d = form._assembler(costs={"cost": form.goals[names[3]]})   # evicts b: the least recently used
    assert len(built) == 4 and len(form._asm_cache) == 3
    assert form._assembler(costs={"cost": form.goals[names[0]]}) is a and len(built) == 4
    assert form._assembler(costs={"cost": form.goals[names[2]]}) is c and len(built) == 4
    assert form._assembler(costs={"cost": form.goals[names[1]]}) is not b and len(built) == 5
    assert d is not None
