"""oracle/admm_oracle.py (the numpy restatement of OSQP's ADMM iteration, the checker of mpcasm_admm)
against what the iteration must do.  osqp itself is absent (a third-party dependency of the reference's
example, biped_mpc_loop.py:13, not vendored, not installed): parity with it is unpinned; what is pinned is
the fixed point -- the KKT conditions of the QP -- and, independently, scipy's SLSQP on the same QPs."""
import numpy as np
import pytest
from scipy.optimize import minimize

from mpcasm import problems
from oracle import admm_oracle as ao
from oracle import qp_oracle as orc


def slsqp(P, q, G, h):
    q, h = q.ravel(), h.ravel()
    r = minimize(lambda x: 0.5 * x @ P @ x + q @ x, np.zeros(P.shape[0]), jac=lambda x: P @ x + q,
                 constraints=[{"type": "ineq", "fun": lambda x: h - G @ x, "jac": lambda x: -G}],
                 method="SLSQP", options={"maxiter": 500, "ftol": 1e-15})
    assert r.status == 0, r.message
    return r.x


def biped_qp(cpu_api, scale):
    form = problems.biped(cpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    given = np.random.default_rng(3).normal(0, scale, [form.given_len, 1])
    G, h, P, q = orc.assemble(form, given)
    return P, q.ravel(), G, h.ravel()


@pytest.mark.parametrize("scale", [0.0, 0.001])
def test_fixed_point_is_the_solution_of_the_bipeds_qp(cpu_api, scale):
    """The walking loop's own QP (biped_mpc_loop.py:50-60; 36 unknowns, 76 limits, 8 of them active):
    iterated to its fixed point the iteration satisfies the KKT conditions to rounding -- Gx <= h,
    y >= 0, y (Gx - h) = 0, Px + q + G'y = 0 -- and lands where SLSQP lands."""
    P, q, G, h = biped_qp(cpu_api, scale)
    x, y, z, (rp, rd) = ao.admm(P, q, G, h, iters=1500, rho=1.0)
    assert rp < 1e-13 and rd < 1e-13
    assert (G @ x - h).max() < 1e-12 and y.min() > -1e-12
    assert np.abs(y * (G @ x - h)).max() < 1e-12
    assert np.abs(P @ x + q + G.T @ y).max() < 1e-12
    assert (y > 1e-9).sum() >= 1                                   # (limits do bind)
    ref = slsqp(P, q, G, h)
    assert np.abs(x - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert 0.5 * x @ P @ x + q @ x <= 0.5 * ref @ P @ ref + q @ ref + 1e-9


def test_random_qps_and_what_a_warm_start_is():
    rng = np.random.default_rng(17)
    for no, nc in ((5, 3), (12, 30), (20, 8)):
        R = rng.standard_normal((no + 3, no))
        P, q = R.T @ R + 0.1 * np.eye(no), rng.standard_normal(no)
        G, h = rng.standard_normal((nc, no)), rng.uniform(0.1, 1.0, nc)         # (x = 0 is strictly feasible)
        x, y, z, res = ao.admm(P, q, G, h, iters=4000, rho=1.0)
        assert max(res) < 1e-11
        # (a convex QP: the KKT conditions are sufficient)
        assert (G @ x - h).max() < 1e-10 and y.min() > -1e-10 and np.abs(y * (G @ x - h)).max() < 1e-10
        assert np.abs(P @ x + q + G.T @ y).max() < 1e-10
        # 30 iterations, then 20 more from where they stopped = 50 in one go
        a = ao.admm(P, q, G, h, iters=50, rho=1.0)
        b = ao.admm(P, q, G, h, iters=30, rho=1.0)
        c = ao.admm(P, q, G, h, *b[:3], iters=20, rho=1.0)
        for u, v in zip(a[:3], c[:3]):
            assert np.array_equal(u, v)
    # the defaults are OSQP's
    assert (ao.RHO, ao.SIGMA, ao.ALPHA) == (0.1, 1e-6, 1.6)
