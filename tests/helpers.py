"""Shared helpers of the tests: golden loading, error measure, tolerances."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: "fp64 P,q,G,h within 1e-10 relative"; measured as
# max |x - ref| / max |ref| over a whole block (an all-zero reference block must be
# reproduced exactly: the error is then max |x| itself).
RTOL = 1e-10
# what the kernels actually achieve on these problems (regression guard)
RTOL_TIGHT = 1e-13


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(x, ref):
    x, ref = np.asarray(x, dtype=float), np.asarray(ref, dtype=float)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = float(np.max(np.abs(ref)))
    err = float(np.max(np.abs(x - ref)))
    return err / scale if scale > 0.0 else err


def assert_close(x, ref, tol=RTOL, what=""):
    err = rel_err(x, ref)
    assert err <= tol, "%s: relative error %.3e > %.1e" % (what, err, tol)
    return err


def ranges_from_json(text):
    return {k: range(v[0], v[1]) for k, v in json.loads(str(text)).items()}


def lti_tracking_problem(api, rng, nx, nu, N, *, scaled=False, extra_unknown=False, given_input=False,
                  two_axis_limit=False, scheduled_cost=False):
    """A random LTI tracking problem (problems.random_lti) with the features the scan form of the
    tiled kernel has to tell apart: a cost on a multiple of a state (coefficient != 1), unknowns that
    are no input of the plant, an input that is GIVEN, a limit over two states (a row of G that is no
    single state row), a cost on part of the horizon (no scan form)."""
    from mpcasm import problems

    A, B = problems.random_lti_matrices(rng, nx, nu)
    inputs = ["u%d" % j for j in range(nu)]
    states = ["s%d" % i for i in range(nx)]
    ext = api.ExtendedSystem.from_cotrol_system(api.ControlSystem(inputs, states, A, B), "x", N)
    form = api.Formulation()
    form.incorporate_dynamics("plant", ext)
    if extra_unknown:
        form.incorporate_dynamics("slack", api.DomainVariable("slack", N))
    if scaled:
        form.incorporate_definition("twice", api.LineCombo({"s1": 2.5}))
    for i, name in enumerate(states):
        var = "twice" if scaled and i == 1 else name
        form.incorporate_goal("track " + name, api.Cost(
            var, float(rng.uniform(0.1, 1)), aim=[float(rng.normal())],
            schedule=range(2, N) if scheduled_cost and i == 0 else range(0)))
    form.incorporate_goal("effort", api.Cost(inputs[-1], 0.3))
    if extra_unknown:
        form.incorporate_goal("slack", api.Cost("slack", 0.2, aim=[0.1]))
    limits = [api.Constraint("s0", 4.0), api.Constraint("s0", 3.0, arrow=[-1]),
              api.Constraint("twice" if scaled else "s1", 2.0, schedule=range(N - 3, N))]
    form.incorporate_constraint("bounds", limits)
    if two_axis_limit:
        form.incorporate_definition("mix", api.LineCombo({"s0": 1.0, "s1": -0.5}))
        form.incorporate_constraint("mixed", [api.Constraint("mix", 1.5)])
    optim = inputs[1:] if given_input else inputs
    form.identify_qp_domain(optim + (["slack"] if extra_unknown else []))
    form.make_preview_matrices()
    return form, A, B
