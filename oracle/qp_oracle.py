"""CPU oracle for the QP-assembly hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference's algorithm for the path
``Formulation.update() -> make_preview_matrices() -> generate_all_qp_matrices()``
plus ``tools.extend_matrices``.  It is the checker the HIP kernels are compared
with.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; nothing under ``mpc-interface_amd/`` does,
and the product path never routes through it.

Parity pinned: every function here is checked against outputs of the real
reference (imported in the build container by ``tests/golden/make_golden.py``)
and against the reference's own fixture ``python/tests/LIP_matrices``; see
``tests/test_oracle_golden.py``.  The LTV generalisation (per-step ``A_k, B_k``)
has no reference counterpart and is pinned only in the degenerate LTI case
("parity unpinned" beyond that).

The functions are duck-typed on the reference's attribute names
(``definitions``, ``of``, ``dynamics``, ``goals``, ``constraints``,
``constraint_boxes``, ``domain`` ...), so they run unchanged on reference
objects (to pin the oracle) and on this repository's mirror classes.

Each function cites the reference lines it follows, relative to
``/root/reference/python/mpc_interface/``.
"""
import numpy as np


# --------------------------------------------------------------------------
# a1  tools.extend_matrices                                   tools.py:14-33
# --------------------------------------------------------------------------
def extend_matrices(N, A, B):
    """``S[k, j, i] = (A^{k+1})[i, j]``, ``U[j][k, l, i] = (A^{k-l} B)[i, j]``.

    Same recurrence as tools.py:23-29 (left-multiply the previous block row by
    ``A``; no repeated squaring), written per block instead of per stacked row.
    """
    A = np.asarray(A, dtype=float)
    B = np.asarray(B, dtype=float)
    n, m = B.shape

    S = np.zeros([N, n, n])
    U = [np.zeros([N, N, n]) for _ in range(m)]

    power = A.copy()                     # A^{k+1}
    blocks = [B.copy()]                  # blocks[d] = A^d B, built by A.dot(previous)
    for k in range(N):
        if k > 0:
            power = A.dot(power)         # tools.py:29
            blocks.append(A.dot(blocks[-1]))   # tools.py:24-26 (column 0 chain)
        S[k] = power.T
        for l in range(k + 1):
            AB = blocks[k - l]
            for j in range(m):
                U[j][k, l, :] = AB[:, j]
    return S, U


def extend_matrices_ltv(N, A_steps, B_steps):
    """Per-step dynamics ``x_{k+1} = A_k x_k + B_k u_k``.

    ``S[k] = (A_k ... A_0)^T``, ``U[j][k, l, :] = (A_k ... A_{l+1} B_l)[:, j]``.
    Generalisation used by BASELINE config C5; equals :func:`extend_matrices`
    when all steps share one ``(A, B)`` (the only case the reference has,
    dynamics.py:222-231).
    """
    A_steps = np.asarray(A_steps, dtype=float)
    B_steps = np.asarray(B_steps, dtype=float)
    n, m = B_steps.shape[1:]
    S = np.zeros([N, n, n])
    U = [np.zeros([N, N, n]) for _ in range(m)]
    row = np.zeros([N, n, m])            # row[l] = A_k..A_{l+1} B_l for the current k
    power = np.eye(n)
    for k in range(N):
        power = A_steps[k].dot(power)
        S[k] = power.T
        for l in range(k):
            row[l] = A_steps[k].dot(row[l])
        row[k] = B_steps[k]
        for l in range(k + 1):
            for j in range(m):
                U[j][k, l, :] = row[l][:, j]
    return S, U


# --------------------------------------------------------------------------
# a3  domain / index maps                                    body.py:97-136
# --------------------------------------------------------------------------
def qp_index_maps(domain, optim_variables):
    """Sizes and ``range`` index maps of the QP unknowns and givens.

    ``domain`` is the ordered ``{variable: size}`` dict; givens are the domain
    variables not listed as unknowns, in domain order (body.py:101-106); IDs
    are prefix sums (body.py:123-133).
    """
    given_variables = [v for v in domain.keys() if v not in optim_variables]

    def ranges(names):
        out, start = {}, 0
        for name in names:
            out[name] = range(start, start + domain[name])
            start += domain[name]
        return out, start

    optim_ID, optim_len = ranges(optim_variables)
    given_ID, given_len = ranges(given_variables)
    return {
        "optim_variables": list(optim_variables),
        "given_variables": given_variables,
        "optim_ID": optim_ID,
        "given_ID": given_ID,
        "optim_len": optim_len,
        "given_len": given_len,
    }


# --------------------------------------------------------------------------
# a4  preview matrices                                      body.py:149-193
# --------------------------------------------------------------------------
def preview_matrices(form, maps=None):
    """``PM[var] = (Mg, Mo)`` for every definition, in definition order."""
    if maps is None:
        maps = qp_index_maps(form.domain, form.optim_variables)
    ng, no = maps["given_len"], maps["optim_len"]
    PM = {}
    for var, combo in form.definitions.items():
        if var in form.of:                                  # body.py:158-177
            rows = form.dynamics[form.of[var]].all_variables[var]
            Mg, Mo = np.zeros([rows, ng]), np.zeros([rows, no])
            for dep, coef in combo.items():
                if dep in maps["given_variables"]:
                    Mg[:, maps["given_ID"][dep]] = coef
                elif dep in maps["optim_variables"]:
                    Mo[:, maps["optim_ID"][dep]] = coef
                else:
                    raise ValueError(
                        "The variable {} in the definition of {} seems to not "
                        "be given nor optimal.".format(dep, var)
                    )
        else:                                               # body.py:179-193
            Mg = Mo = None
            for dep, coef in combo.items():
                c = np.array(coef)
                tg, to = c.dot(PM[dep][0]), c.dot(PM[dep][1])
                if Mg is None:
                    Mg, Mo = tg, to
                    if Mg.ndim == 1:
                        Mg, Mo = Mg[None, :], Mo[None, :]
                else:
                    Mg += tg
                    Mo += to
        PM[var] = (Mg, Mo)
    return PM


def arrange_given(maps, collector):
    """Scatter the collector's column vectors by ``given_ID`` (body.py:195-207)."""
    if not maps["given_len"]:
        return np.array([])
    given = np.zeros([maps["given_len"], 1])
    for var, ids in maps["given_ID"].items():
        given[ids] = collector[var]
    return given


# --------------------------------------------------------------------------
# a6  constraint row rule, coefficients and bound    restrictions.py:147-199
# --------------------------------------------------------------------------
def constraint_nlines(limit):
    if limit.L:
        return limit.L[0].shape[0]
    if limit.schedule:
        return limit.schedule.stop - limit.schedule.start
    rows = [limit.arrow.shape[0], limit.center.shape[0], limit.extreme.shape[0]]
    wide = [r for r in rows if r != 1]
    return wide[0] if wide else None


def constraint_coefficients(limit):
    cols = [limit.arrow[:, i][:, None] for i in range(len(limit.axes))]
    if limit.L:
        return [c * l for c, l in zip(cols, limit.L)]
    return cols


def constraint_bound(limit):
    return limit.extreme + np.sum(limit.arrow * limit.center, axis=1).reshape([-1, 1])


# --------------------------------------------------------------------------
# a7  one constraint                                        body.py:236-264
# --------------------------------------------------------------------------
def qp_constraint(PM, limit, given):
    Mg0 = PM[limit.variable + limit.axes[0]][0]
    rows = Mg0.shape[0]
    nlines = constraint_nlines(limit)
    out_rows = rows if nlines is None else nlines
    ng = Mg0.shape[1]
    no = PM[limit.variable + limit.axes[0]][1].shape[1]

    cMg, cMo = np.zeros([out_rows, ng]), np.zeros([out_rows, no])
    picked = limit.schedule if limit.schedule else range(rows)
    coefs = constraint_coefficients(limit)
    for i, axis in enumerate(limit.axes):
        Mg, Mo = PM[limit.variable + axis]
        if limit.L:
            cMg += coefs[i] @ Mg[picked]
            cMo += coefs[i] @ Mo[picked]
        else:
            cMg += coefs[i] * Mg[picked]
            cMo += coefs[i] * Mo[picked]
    return cMo, constraint_bound(limit) - cMg @ given


# --------------------------------------------------------------------------
# a9  one cost                                              body.py:266-302
# --------------------------------------------------------------------------
def qp_cost(PM, cost, given):
    first = PM[cost.variable + cost.axes[0]]
    rows, no = first[0].shape[0], first[1].shape[1]
    Q, q = np.zeros([no, no]), np.zeros([no, 1])
    picked = cost.schedule if cost.schedule else range(rows)

    for i, axis in enumerate(cost.axes):
        vMg, vMo = (M[picked] for M in PM[cost.variable + axis])
        cMg, cMo = (M[picked] for M in PM[cost.cross + axis])
        if cost.L:
            vMg, vMo = cost.L[i] @ vMg, cost.L[i] @ vMo
        if cost.cross_L:
            cMg, cMo = cost.cross_L[i] @ cMg, cost.cross_L[i] @ cMo

        Q += cost.weight * vMo.T @ cMo                       # (w vMo^T) @ cMo
        q += (
            cost.weight
            * (vMo.T @ (cMg @ given - cost.cross_aim[:, i])
               + cMo.T @ (vMg @ given - cost.aim[:, i]))
            / 2
        )
    return Q, q


# --------------------------------------------------------------------------
# a8, a10, a11  stacking                                    body.py:304-348
# --------------------------------------------------------------------------
def all_limits(form):
    """Constraint order of the stacked G: named constraints, then boxes."""
    limits = [l for group in form.constraints.values() for l in group]
    limits += [l for box in form.constraint_boxes.values() for l in box.constraints]
    return limits


def qp_all_constraints(form, PM, given):
    parts = [qp_constraint(PM, limit, given) for limit in all_limits(form)]
    return np.vstack([p[0] for p in parts]), np.vstack([p[1] for p in parts])


def qp_all_costs(form, PM, given):
    parts = [qp_cost(PM, cost, given) for cost in form.goals.values()]
    return (np.add.reduce([p[0] for p in parts]),
            np.add.reduce([p[1] for p in parts]))


def assemble(form, given, PM=None, maps=None):
    """``(A, h, Q, q)`` = qpsolvers ``(G, h, P, q)`` (body.py:333-348)."""
    if maps is None:
        maps = qp_index_maps(form.domain, form.optim_variables)
    if PM is None:
        PM = preview_matrices(form, maps)
    A, h = qp_all_constraints(form, PM, given)
    Q, q = qp_all_costs(form, PM, given)
    return A, h, Q, q


def preview(PM, given, optim, variable, axes=None):
    """``Mg @ given + Mo @ optim`` (body.py:209-219)."""
    if axes is None:
        Mg, Mo = PM[variable]
        return Mg @ given + Mo @ optim
    return np.hstack(
        [PM[variable + a][0] @ given + PM[variable + a][1] @ optim for a in axes]
    )
