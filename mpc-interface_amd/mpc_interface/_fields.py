"""Field handling shared by the problem-description holders (Cost, Constraint): how an ``L``
argument becomes one matrix per axis and how ``update()`` replaces numeric fields.  Host-side
data plumbing only; the numbers are consumed by the plan compiler / the kernels."""
import numpy as np


def per_axis_matrices(L, axes_len, columns):
    """``(matrices, problem)``: ``L`` as a list with one matrix per axis (a single matrix is
    shared by all axes; an empty list means no L) and, instead of raising, the exception that
    describes what is wrong with it -- the callers decide what to store before raising.
    ``columns``: what the schedule demands of every matrix' last dimension (falsy: anything).
    Same rules and messages as the reference (goal.py:88-104, restrictions.py:75-90)."""
    matrices = list(L) if isinstance(L, list) else [L]
    if len(matrices) == 1:
        matrices = matrices * axes_len
    problem = None
    if len(matrices) not in (axes_len, 0):
        problem = IndexError(
            "'L' must have 0, 1 or len(axes) = {} elements".format(axes_len))
    elif columns and any(np.shape(m)[-1] != columns for m in matrices):
        problem = ValueError(
            "arrays in L must have {} columns, which is given by the "
            "'schedule'.".format(columns))
    return matrices, problem


def as_rows(value, columns):
    """A numeric field as the 2-D array the holders keep: one row per line."""
    return np.array(value).reshape([-1, columns])
