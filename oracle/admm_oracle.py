"""numpy restatement of OSQP's ADMM iteration  --  TEST INFRASTRUCTURE ONLY.

The reference's walking loop hands every assembled QP to ``qpsolvers.osqp_solve_qp(P, q, G, h)``
(use_examples/simple_functional_example/biped_mpc_loop.py:57-60).  ``qpsolvers`` and ``osqp`` are
third-party dependencies that are NOT vendored in /root/reference and not installed here (the reference
names them without a version: README.md:33-37, ``pip3 install qpsolvers osqp``), so this file restates
the published algorithm -- B. Stellato, G. Banjac, P. Goulart, A. Bemporad, S. Boyd, "OSQP: an operator
splitting solver for quadratic programs", Math. Prog. Comp. 12 (2020), Algorithm 1 -- for the problem
the loop poses, ``min 1/2 x'Px + q'x  s.t.  Gx <= h`` (l = -inf, u = h, no equalities, body.py:331):

    solve (P + sigma I + rho G'G) xt = sigma x - q + G'(rho z - y)      (the reduced KKT system, eq. (25))
    zt = G xt
    x+ = alpha xt + (1 - alpha) x
    z+ = min(alpha zt + (1 - alpha) z + y / rho, h)                     (the projection onto (-inf, h])
    y+ = y + rho (alpha zt + (1 - alpha) z - z+)

with OSQP's default steps rho = 0.1, sigma = 1e-6, alpha = 1.6 -- the plain iteration: no problem
scaling, no adaptive rho, no polishing (those change the path of the iterates, not its fixed point).
**Parity unpinned against osqp itself** (absent); pinned instead on what the iteration must do:
its fixed point satisfies the KKT conditions of the QP, checked against scipy's SLSQP on the biped's
own QPs (tests/test_admm_oracle.py).  The device kernel (mpcasm_admm) is held to these iterates.
"""
import numpy as np

RHO, SIGMA, ALPHA = 0.1, 1e-6, 1.6


def admm(P, q, G, h, x=None, y=None, z=None, iters=50, rho=RHO, sigma=SIGMA, alpha=ALPHA):
    """``iters`` iterations from ``(x, y, z)`` (zeros when None; ``z`` = min(Gx, h) when only it is
    None).  Returns ``x, y, z`` and the residuals ``(|Gx - z|_inf, |Px + q + G'y|_inf)``."""
    P, G = np.asarray(P, dtype=np.float64), np.asarray(G, dtype=np.float64)
    q, h = np.asarray(q, dtype=np.float64).ravel(), np.asarray(h, dtype=np.float64).ravel()
    no, nc = P.shape[0], G.shape[0]
    x = np.zeros(no) if x is None else np.array(x, dtype=np.float64).ravel()
    y = np.zeros(nc) if y is None else np.array(y, dtype=np.float64).ravel()
    z = np.minimum(G @ x, h) if z is None else np.array(z, dtype=np.float64).ravel()
    M = P + sigma * np.eye(no) + rho * (G.T @ G)
    for _ in range(iters):
        xt = np.linalg.solve(M, sigma * x - q + G.T @ (rho * z - y))
        zt = G @ xt
        x = alpha * xt + (1.0 - alpha) * x
        zr = alpha * zt + (1.0 - alpha) * z
        zn = np.minimum(zr + y / rho, h)
        y = y + rho * (zr - zn)
        z = zn
    return x, y, z, residuals(P, q, G, x, y, z)


def residuals(P, q, G, x, y, z):
    """OSQP's primal and dual residuals (eq. (23), infinity norms)."""
    return float(np.abs(G @ x - z).max(initial=0.0)), float(np.abs(P @ x + q + G.T @ y).max(initial=0.0))
