"""Numeric helpers of the problem description (host side) and the K1 entry point.

Mirror of /root/reference/python/mpc_interface/tools.py.  Only one function is
on the accelerated path: :func:`extend_matrices` (reference tools.py:14-33),
which here calls the HIP Toeplitz-fill kernel through the C-ABI
(``mpcasm_fill_su``, include/mpcasm.h).  Everything else is the small per-tick
host logic that decides problem *structure* (step plan, sizes, box centres) and
stays on the host exactly as in the reference.

The named LIPM systems are discretised in closed form (the reference derives
the same matrices symbolically with sympy at set-up, tools.py:240-358).
"""
import math

import numpy as np


# --------------------------------------------------------------------------
# K1: horizon extension  x_{k+1} = A x_k + B u_k  ->  S, U      (HIP, C-ABI)
# --------------------------------------------------------------------------
def extend_matrices(N, A, B):
    """Prediction matrices over a horizon of ``N`` samples (tools.py:14-33).

    Returns ``S`` with shape ``(N, n, n)`` and a list of ``m`` arrays ``U[j]``
    with shape ``(N, N, n)`` such that ``S[k, j, i] = (A^{k+1})[i, j]`` and
    ``U[j][k, l, i] = (A^{k-l} B)[i, j]`` for ``l <= k`` (zero above).

    Computed on the GPU by ``mpcasm_fill_su``; raises ``RuntimeError`` when the
    HIP library or a device is missing (there is no CPU fallback).
    """
    from mpcasm import engine as _engine

    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    n, m = B.shape
    S, U = _engine.fill_su_numpy(A.reshape(1, n, n), B.reshape(1, n, m), int(N))
    return S[0], [U[0, j] for j in range(m)]


# --------------------------------------------------------------------------
# step planning (structure decisions, host)
# --------------------------------------------------------------------------
def _steps_in_preview(step_times, count, N):
    """Step instants inside ``[count, count + N - 1)`` (tools.py:84, :145)."""
    step_times = np.asarray(step_times)
    inside = (step_times >= count) * (step_times < count + N - 1)
    return step_times[inside]


def plan_steps(N, count=0, step_times=None, regular_time=None, phase=0):
    """Integer step-indicator matrix ``E`` (N x p), tools.py:79-101.

    ``E[k, s] = 1`` when preview sample ``count + k`` lies strictly after the
    ``s``-th step instant of the preview window.
    """
    preview_times = count + np.arange(N)

    if step_times is not None:
        next_steps = _steps_in_preview(step_times, count, N)
    elif regular_time is not None:
        last = count + N - 1
        next_steps = np.array(
            [t for t in preview_times
             if (t + phase) % regular_time == 0 and t < last]
        )
    else:
        raise KeyError(
            "either the step_times or some regular_time for steps must be provided"
        )

    return (preview_times.reshape([N, 1]) > next_steps).astype(int)


def update_step_matrices(extSyst, **kargs):
    """Re-plan the steps of a step system (tools.py:36-76).

    Needs ``step_times`` or ``regular_time``; optional ``count`` and ``w_phase``.
    """
    N = extSyst.matrices[-1].shape[0]
    count = kargs.get("count", 0)

    if "step_times" in kargs:
        step_times, regular_time = kargs["step_times"], None
    elif "regular_time" in kargs:
        step_times, regular_time = None, kargs["regular_time"]
    else:
        raise KeyError(
            "This funtion needs either 'step_times' or 'regular_time', "
            "but the kargs introduced are {}".format(kargs.keys())
        )

    E = plan_steps(N, count, step_times, regular_time, kargs.get("w_phase", 0))
    extSyst.matrices[0] = E[:, :, None]


def n_predicted_steps(count, N, step_times):
    """Number of steps inside the preview window (tools.py:144-146)."""
    return _steps_in_preview(step_times, count, N).size


def count_yawls(domVar, **kargs):
    """tools.py:108-112."""
    domVar.domain["yawl"] = n_predicted_steps(
        kargs.get("count", 0), kargs["N"], kargs["step_times"]
    )


def step_average_velocity(count, N, step_times):
    """Finite-difference operator between consecutive steps (tools.py:115-129)."""
    preview_times = count + np.arange(N)
    marks = np.hstack([_steps_in_preview(step_times, count, N), count + N - 1])
    begin = (marks[:-1].reshape([-1, 1]) == preview_times).astype(int)
    end = (marks[1:].reshape([-1, 1]) == preview_times).astype(int)
    L = (end - begin) / (marks[1:] - marks[:-1])[:, None]
    return np.hstack([L, L])


def linear_Rotations(old_yawls):
    """Linearised rotation of the steps (tools.py:132-141)."""
    s, c = np.sin(old_yawls), np.cos(old_yawls)
    L = np.vstack([np.diag(-s), np.diag(c)])
    return L, L @ old_yawls[:, None] - np.vstack([c[:, None], s[:, None]])


def find_step_centers(step_count, n_next_steps, xy_lenght):
    """Alternating left/right centres of the next stepping areas (tools.py:158-165)."""
    side = (-1) ** (step_count + 1)
    sign = np.tile([[1], [-1]], [n_next_steps // 2 + 1, 1])[:n_next_steps]
    return np.hstack(
        [np.ones([n_next_steps, 1]) * xy_lenght[0], side * sign * xy_lenght[1]]
    )


def update_stepping_area(box, **kargs):
    """tools.py:149-155."""
    centers = find_step_centers(
        kargs["step_count"], kargs["n_next_steps"], kargs["xy_lenght"]
    )
    for limit in box.constraints:
        limit.update(center=centers)


def adapt_size(stamps, **kargs):
    """Resize the 'stamps' output to the current number of steps (tools.py:204-215)."""
    dynamics = kargs["extSyst"]
    axis = stamps.variables[0][2:]
    p = dynamics.domain["Ds" + axis]
    stamps.matrices[stamps.variables.index("Ds" + axis)] = np.tril(
        np.ones([p + 1, p]), -1
    )
    stamps.matrices[stamps.variables.index("s0" + axis)] = np.ones([p + 1, 1])


def reduce_by_time(box, **kargs):
    """Shrink a swing-foot box as the step advances (tools.py:168-182)."""
    box.recenter_in_TS(kargs["current_swing_pose"].get_translation()[:2])
    usable = kargs["step_duration"] - kargs["landing_advance"]
    s = (usable - kargs["current_ss_time"]) / usable
    box.scale_box(s if s > 0 else 1e-2)


def recenter_on_real_state_x(box, **kargs):
    box.recenter_in_SS(new_center=kargs["x0_x"])


def recenter_on_real_state_y(box, **kargs):
    box.recenter_in_SS(new_center=kargs["x0_y"])


def recenter_support(box, **kargs):
    box.recenter_in_TS(new_center=kargs["s0"])


def make_simetric_vertices(xy_corner):
    """Four corners of an axis-aligned rectangle (tools.py:197-201)."""
    sx = np.array([1, -1, -1, 1])[:, None] * xy_corner[0]
    sy = np.array([1, 1, -1, -1])[:, None] * xy_corner[1]
    return np.hstack([sx, sy])


def rotation2D(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s], [s, c]])


def rotation3D(angle, axis="z"):
    """tools.py:222-237."""
    r2 = rotation2D(angle)
    R = np.zeros([3, 3])
    if axis == "x":
        R[0, 0] = 1
        R[1:3, 1:3] = r2
    elif axis == "y":
        R[1, 1] = 1
        R[[0, 2, 0, 2], [0, 0, 2, 2]] = r2.flatten()
    elif axis == "z":
        R[2, 2] = 1
        R[0:2, 0:2] = r2
    return R


# --------------------------------------------------------------------------
# named linear inverted pendulum systems (set-up time, closed form)
# --------------------------------------------------------------------------
_SYSTEM_VARIABLES = {
    # name: (inputs, states)                       reference tools.py:288-325
    "P->CC": (["cCoP"], ["CoM", "CoM_dot"]),
    "P->X": (["cCoP"], ["DCM"]),
    "dP->CCC": (["cCoP_dot"], ["CoM", "CoM_dot", "CoM_ddot"]),
    "dP->CCP": (["cCoP_dot"], ["CoM", "CoM_dot", "cCoP"]),
    "J->CCC": (["CoM_dddot"], ["CoM", "CoM_dot", "CoM_ddot"]),
}


def get_system_variables(system):
    """Input and state variable names of a named system (tools.py:288-325)."""
    inputs, states = _SYSTEM_VARIABLES[system]
    return list(inputs), list(states)


def _hyp(tau, omega):
    x = omega * tau
    return math.cosh(x), math.sinh(x)


def _lipm_closed_form(system):
    """Exact zero-order-hold discretisation ``A = exp(G tau)``,
    ``B = int_0^tau exp(G t) dt H`` of the continuous models of tools.py:255-278."""
    if system == "P->CC":
        def get_A(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([[ch, sh / omega], [omega * sh, ch]])

        def get_B(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([[1 - ch], [-omega * sh]])

    elif system == "P->X":
        def get_A(tau=None, omega=None, **kwargs):
            return np.array([[math.exp(omega * tau)]])

        def get_B(tau=None, omega=None, **kwargs):
            return np.array([[1 - math.exp(omega * tau)]])

    elif system == "dP->CCC":
        def get_A(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([
                [1, sh / omega, (ch - 1) / omega**2],
                [0, ch, sh / omega],
                [0, omega * sh, ch],
            ])

        def get_B(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([[tau - sh / omega], [1 - ch], [-omega * sh]])

    elif system == "dP->CCP":
        def get_A(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([
                [ch, sh / omega, 1 - ch],
                [omega * sh, ch, -omega * sh],
                [0, 0, 1],
            ])

        def get_B(tau=None, omega=None, **kwargs):
            ch, sh = _hyp(tau, omega)
            return np.array([[tau - sh / omega], [1 - ch], [tau]])

    elif system == "J->CCC":
        def get_A(tau=None, **kwargs):
            return np.array([[1, tau, tau**2 / 2], [0, 1, tau], [0, 0, 1.0]])

        def get_B(tau=None, **kwargs):
            return np.array([[tau**3 / 6], [tau**2 / 2], [tau * 1.0]])

    else:
        raise KeyError("unknown system formulation '{}'".format(system))

    return get_A, get_B


def get_system_matrices(system):
    """``(get_A, get_B, parameters)`` of a named system (tools.py:240-285).

    ``get_A(tau=..., omega=...)`` and ``get_B`` return the discrete matrices
    ``(n, n)`` and ``(n, 1)``.
    """
    get_A, get_B = _lipm_closed_form(system)
    if system == "J->CCC":
        parameters = ("tau=None", "**kwargs")
    else:
        parameters = ("tau=None", "omega=None", "**kwargs")
    return get_A, get_B, parameters


def discretize(G, *H, tau=None):
    """Exact discretisation of ``dx/dt = G x + h1 u1 + ...`` (tools.py:328-358).

    Numeric counterpart of the reference's symbolic routine: returns
    ``(A, b1, b2, ...)`` for sampling period ``tau`` using one matrix
    exponential of the augmented system.
    """
    from scipy.linalg import expm

    if tau is None:
        raise ValueError("a numeric sampling period 'tau' is required")
    G = np.atleast_2d(np.asarray(G, dtype=float))
    n = G.shape[0]
    cols = [np.asarray(h, dtype=float).reshape(n, -1) for h in H]
    Hm = np.hstack(cols) if cols else np.zeros([n, 0])
    aug = np.zeros([n + Hm.shape[1]] * 2)
    aug[:n, :n] = G
    aug[:n, n:] = Hm
    E = expm(aug * tau)
    out, c0 = [E[:n, :n]], n
    for c in cols:
        out.append(E[:n, c0:c0 + c.shape[1]])
        c0 += c.shape[1]
    return tuple(out)


def do_not_update(sys, **kargs):
    return None
