#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/*.npz from the REAL reference.

Run in the build container only (the reference lives at /root/reference and
never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_golden.py

What it does
  * imports the reference package read-only and builds each problem with the
    builders of ``mpcasm.problems`` (the same construction code the tests later
    run on this repository's own ``mpc_interface`` mirror);
  * records inputs and the reference's outputs (index maps, preview matrices,
    per-constraint / per-cost blocks, stacked A, h, Q, q; S, U of
    ``extend_matrices``; facet arrays of ``Box``; named-system matrices);
  * pins the oracle: runs ``oracle/qp_oracle.py`` on the very same reference
    objects and asserts agreement before anything is written.

The fixtures are data only (numbers and variable names); no reference source
text is stored.  Sets G1..G5 follow SURVEY.md section 8c.
"""
import json
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/python"

sys.dont_write_bytecode = True
sys.path.insert(0, REF)                                   # the reference package
sys.path.insert(1, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(2, ROOT)

# only the builders are taken from this repository; "mpc_interface" resolves
# to the reference because REF comes first on sys.path
import mpc_interface                                       # noqa: E402
assert mpc_interface.__file__.startswith(REF), mpc_interface.__file__
from mpcasm import problems                                # noqa: E402
from oracle import qp_oracle as orc                        # noqa: E402

api = problems.load_api("mpc_interface")
TOL = 1e-12


def close(a, b, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.max(np.abs(b))) if b.size else 1.0)
    err = float(np.max(np.abs(a - b))) / scale if b.size else 0.0
    assert err <= TOL, (what, err)
    return err


def ranges_json(ids):
    return json.dumps({k: [r.start, r.stop] for k, r in ids.items()})


def snapshot(form, given, out, prefix, with_pm=True, with_parts=True):
    """Record the reference's results for one formulation state and check the
    oracle against them."""
    maps = orc.qp_index_maps(form.domain, form.optim_variables)
    assert maps["optim_ID"] == {k: form.optim_ID[k] for k in maps["optim_ID"]}
    assert maps["given_ID"] == {k: form.given_ID[k] for k in maps["given_ID"]}
    assert maps["optim_len"] == form.optim_len and maps["given_len"] == form.given_len
    out[prefix + "optim_ID"] = ranges_json(
        {k: form.optim_ID[k] for k in form.optim_variables})
    out[prefix + "given_ID"] = ranges_json(
        {k: form.given_ID[k] for k in form.given_variables})
    out[prefix + "given"] = given

    PM = orc.preview_matrices(form, maps)
    names = list(form.PM.keys())
    assert names == list(PM.keys())
    for var in names:
        close(PM[var][0], form.PM[var][0], prefix + var + " Mg")
        close(PM[var][1], form.PM[var][1], prefix + var + " Mo")
        if with_pm:
            out[prefix + "PM/" + var + "/Mg"] = form.PM[var][0]
            out[prefix + "PM/" + var + "/Mo"] = form.PM[var][1]
    out[prefix + "definitions"] = json.dumps(names)

    limits = orc.all_limits(form)
    for k, limit in enumerate(limits):
        A, h = form.generate_qp_constraint(limit, given)
        Ao, ho = orc.qp_constraint(PM, limit, given)
        close(Ao, A, prefix + "limit A %d" % k)
        close(ho, h, prefix + "limit h %d" % k)
        assert orc.constraint_nlines(limit) == limit.nlines
        close(orc.constraint_bound(limit), limit.bound(), "bound")
        if with_parts:
            out[prefix + "limit%d/A" % k], out[prefix + "limit%d/h" % k] = A, h
    for name, cost in form.goals.items():
        Q, q = form.generate_qp_cost(cost, given)
        Qo, qo = orc.qp_cost(PM, cost, given)
        close(Qo, Q, prefix + "cost Q " + name)
        close(qo, q, prefix + "cost q " + name)
        if with_parts:
            out[prefix + "cost/" + name + "/Q"], out[prefix + "cost/" + name + "/q"] = Q, q

    A, h, Q, q = form.generate_all_qp_matrices(given)
    Ao, ho, Qo, qo = orc.assemble(form, given, PM, maps)
    for mine, ref, nm in ((Ao, A, "A"), (ho, h, "h"), (Qo, Q, "Q"), (qo, q, "q")):
        close(mine, ref, prefix + nm)
    out[prefix + "A"], out[prefix + "h"] = A, h
    out[prefix + "Q"], out[prefix + "q"] = Q, q


def save(name, out):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB  %d arrays" % (name + ".npz", os.path.getsize(path) / 1024, len(out)))


# --------------------------------------------------------------------------
# G1  extend_matrices
# --------------------------------------------------------------------------
def g1_extend():
    out = {}
    with open(os.path.join(REF, "tests", "LIP_matrices"), "rb") as f:
        saved = pickle.load(f)["matrices"]
    lip = api.ControlSystem.from_name(system_name="J->CCC", tau=0.1, omega=3.3445,
                                      axes=["_x", "_y"])
    ext = api.ExtendedSystem.from_cotrol_system(lip, state_vector_name="x", horizon_lenght=36)
    assert np.isclose(ext.matrices[0], saved[0]).all()
    assert np.isclose(ext.matrices[1], saved[1]).all()
    out["lip36/A"], out["lip36/B"] = lip.A, lip.B
    out["lip36/U0"], out["lip36/S"] = saved[0], saved[1]      # the reference's own fixture
    S, U = orc.extend_matrices(36, lip.A, lip.B)
    close(S, saved[1], "lip36 S")
    close(U[0], saved[0], "lip36 U")

    rng = np.random.default_rng(20259)
    for n, m, N in [(3, 1, 16), (3, 1, 32), (3, 1, 100), (8, 6, 20), (12, 6, 64), (1, 1, 5),
                    (2, 3, 1)]:
        A = rng.standard_normal((n, n)) / np.sqrt(n)
        B = rng.standard_normal((n, m))
        S, U = api.tools.extend_matrices(N, A, B)
        So, Uo = orc.extend_matrices(N, A, B)
        close(So, S, "S")
        for j in range(m):
            close(Uo[j], U[j], "U")
        Sl, Ul = orc.extend_matrices_ltv(N, np.stack([A] * N), np.stack([B] * N))
        close(Sl, S, "S ltv")
        for j in range(m):
            close(Ul[j], U[j], "U ltv")
        key = "lti_n%d_m%d_N%d/" % (n, m, N)
        out[key + "A"], out[key + "B"], out[key + "S"] = A, B, S
        if U[0].size * m <= 40000:
            out[key + "U"] = np.stack(U)
        else:   # keep the fixture small: a few block rows of the first and last input
            rows = np.array([0, 1, N // 2, N - 1])
            out[key + "U_rows"] = rows
            out[key + "U_first"] = U[0][rows]
            out[key + "U_last"] = U[m - 1][rows]
            out[key + "U_sums"] = np.array([u.sum() for u in U])
            out[key + "U_abs_sums"] = np.array([np.abs(u).sum() for u in U])
    save("g1_extend", out)


# --------------------------------------------------------------------------
# G2  the formulation of the reference's own test_body.py (every branch of
#     generate_qp_constraint / generate_qp_cost: L / no L, schedule, cross, box)
# --------------------------------------------------------------------------
def g2_body():
    out = {}
    form = problems.body_case(api)
    given = form.arrange_given(
        {v: np.array(list(r)).reshape([-1, 1]) for v, r in form.given_ID.items()})
    snapshot(form, given, out, "arange/")
    rng = np.random.default_rng(20258)
    given = rng.standard_normal([form.given_len, 1])
    snapshot(form, given, out, "random/", with_pm=False)
    save("g2_body", out)


# --------------------------------------------------------------------------
# G3  biped walking ticks (both QP widths), N=16 and N=24
# --------------------------------------------------------------------------
EXAMPLE = "/root/reference/python/use_examples/simple_functional_example"


def reference_example(step_samples):
    """The reference's OWN example formulation, ``formulate_biped(conf)`` of
    use_examples/simple_functional_example/biped_formulation.py:23-191, on its
    biped_configuration with ``step_samples`` overridden (8: N = 16, BASELINE configs C1 / C2;
    12: the example as shipped, N = 24).  Imported read-only, never written into a fixture."""
    import types

    if EXAMPLE not in sys.path:
        sys.path.insert(3, EXAMPLE)
    import biped_configuration
    import biped_formulation

    conf = types.SimpleNamespace(**{k: v for k, v in vars(biped_configuration).items()
                                    if not k.startswith("_")})
    conf.step_samples = step_samples
    conf.horizon_lenght = conf.num_steps * step_samples
    return biped_formulation.formulate_biped(conf), conf


def same_problem(form, ref_form, given, where):
    """``problems.biped`` builds the very problem of the reference's example: identical index
    maps and (A, h, Q, q), bit for bit."""
    assert list(form.optim_ID.items()) == list(ref_form.optim_ID.items()), where
    assert list(form.given_ID.items()) == list(ref_form.given_ID.items()), where
    mine = form.generate_all_qp_matrices(given)
    theirs = ref_form.generate_all_qp_matrices(given)
    for a, b, name in zip(mine, theirs, "AhQq"):
        assert a.shape == b.shape and np.array_equal(a, b), (where, name)


def g3_biped():
    for step_samples, keep in ((8, (0, 1, 6, 7, 8, 9, 15, 17)), (12, (0, 10, 11, 12, 17))):
        out = {}
        conf = problems.BipedConfig(step_samples=step_samples)
        form = problems.biped(api, conf)
        ref_form, ref_conf = reference_example(step_samples)
        assert ref_conf.horizon_lenght == conf.horizon_lenght
        clock = problems.StepClock(conf.step_samples, form.domain["Ds_x"])
        rng = np.random.default_rng(20260 + step_samples)
        shapes = []
        for tick in range(18):
            form.update(step_times=clock.step_times, step_count=clock.step_count)
            ref_form.update(step_times=clock.step_times, step_count=clock.step_count)
            collector = problems.biped_given_collector(form, rng, bias_sigma=0.01)
            given = form.arrange_given(collector)
            same_problem(form, ref_form, given, "biped N=%d tick %d" % (conf.horizon_lenght, tick))
            if tick in keep:
                p = "tick%02d/" % tick
                out[p + "step_times"] = clock.step_times.copy()
                out[p + "step_count"] = np.array(clock.step_count)
                first = tick == keep[0]
                snapshot(form, given, out, p, with_pm=first, with_parts=first)
                box = form.constraint_boxes["stepping area"]
                out[p + "stepping_centers"] = np.stack([l.center for l in box.constraints])
            A, h, Q, q = form.generate_all_qp_matrices(given)
            shapes.append([tick, clock.step_count, Q.shape[0], A.shape[0]])
            clock.tick()
        out["shapes"] = np.array(shapes)
        out["ticks"] = np.array(keep)
        print("biped N=%d shapes (tick, step_count, no, nc):" % conf.horizon_lenght,
              [tuple(s) for s in shapes])
        save("g3_biped_N%d" % conf.horizon_lenght, out)


# --------------------------------------------------------------------------
# G4  Constraint / Box geometry
# --------------------------------------------------------------------------
def g4_geometry():
    out = {}
    t = api.tools
    boxes = {
        "foot2d": api.Box.task_space("v", t.make_simetric_vertices(np.array([0.1, 0.05])),
                                     ["_x", "_y"]),
        "stepping2d": api.Box.task_space("v", t.make_simetric_vertices(np.array([0.3, 0.1])),
                                         ["_x", "_y"]),
        "diamond2d": api.Box.task_space("v", np.array([[0, 1], [1, 0], [0, -1], [-1, 0]]),
                                        ["_x", "_y"]),
        "cuboid3d": api.Box.task_space(
            "v", np.array([[sx * 0.2, sy * 0.1, sz * 0.05]
                           for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)]) + 0.3,
            ["_x", "_y", "_z"]),
        "segment1d": api.Box.task_space("v", np.array([[-0.2], [0.5]])),
        "state_space": api.Box.state_space(
            "v", np.array([[0.0, 1], [1, 0.5], [0.2, -1], [-1, 0]]), schedule=range(0, 2)),
    }
    boxes["foot2d_margin"] = api.Box.task_space(
        "v", t.make_simetric_vertices(np.array([0.1, 0.05])), ["_x", "_y"])
    boxes["foot2d_margin"].set_safety_margin(0.02)
    for name, box in boxes.items():
        out[name + "/n"] = np.array(len(box.constraints))
        for k, limit in enumerate(box.constraints):
            p = "%s/%d/" % (name, k)
            out[p + "arrow"], out[p + "center"] = limit.arrow, limit.center
            out[p + "extreme"], out[p + "bound"] = limit.extreme, limit.bound()
            if limit.L:
                out[p + "L"] = np.stack(limit.L)
    # the bound() values the reference's own test pins (test_restrictions.py:99-104)
    c = api.Constraint("CoM", [2, 3], axes=["_x", "_y"], arrow=[[1, 0], [0, -1]],
                       center=[1, 1])
    out["twoaxes/bound"] = c.bound()
    c2 = api.Constraint("CoM", [-2, 3], axes=["_x", "_y"], arrow=[[1, 0], [0, -1]])
    out["flipped/arrow"], out["flipped/extreme"] = c2.arrow, c2.extreme
    save("g4_geometry", out)


# --------------------------------------------------------------------------
# G5  named systems
# --------------------------------------------------------------------------
def g5_systems():
    out = {}
    for name in ("P->CC", "P->X", "dP->CCC", "dP->CCP", "J->CCC"):
        get_A, get_B, _ = api.tools.get_system_matrices(name)
        for tau, omega in ((0.1, 3.5), (0.1, 3.3445), (0.05, 3.0)):
            key = "%s/tau%g_omega%g/" % (name.replace("->", "_to_"), tau, omega)
            out[key + "A"] = np.asarray(get_A(tau=tau, omega=omega), dtype=float)
            out[key + "B"] = np.asarray(get_B(tau=tau, omega=omega), dtype=float)
    save("g5_systems", out)


# --------------------------------------------------------------------------
# G6  builder-defined larger configs (C3, reduced C4) -- value pins for shapes
#     the biped does not reach (3 axes, per-row constraints on plain states)
# --------------------------------------------------------------------------
def g6_configs():
    out = {}
    form = problems.lipm3d(api, N=32)
    rng = np.random.default_rng(20261)
    given = rng.normal(0, 0.05, [form.given_len, 1])
    snapshot(form, given, out, "lipm3d_N32/", with_pm=False, with_parts=False)

    form = problems.random_lti(api, np.random.default_rng(20262), nx=12, nu=6, N=8)
    given = np.random.default_rng(1).standard_normal([form.given_len, 1])
    snapshot(form, given, out, "lti_nx12_nu6_N8/", with_pm=False, with_parts=False)
    save("g6_configs", out)


if __name__ == "__main__":
    g1_extend()
    g2_body()
    g3_biped()
    g4_geometry()
    g5_systems()
    g6_configs()
    print("oracle pinned against the reference on every case above (tol %g)" % TOL)
