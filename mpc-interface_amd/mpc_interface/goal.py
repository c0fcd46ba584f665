"""Cost description.

Host-side mirror of /root/reference/python/mpc_interface/goal.py:11-161.  A
``Cost`` is a pure data holder; its numeric fields (weight, aim, cross_aim)
become per-instance device parameters of the batched assembly
(``mpcasm.plan``), its ``L``/``schedule`` become row-set structure.
"""
import numpy as np

from ._fields import per_axis_matrices


class Cost:
    """Bring ``V = L @ v[schedule]`` towards ``aim`` with some ``weight``.

    With ``cross`` the product ``(V - aim) * (C - cross_aim)`` is penalised
    instead, ``C = cross_L @ c[schedule]`` (goal.py:24-60).  Without it the
    cross fields alias the plain ones (``cross_aim is aim``, ``cross_L is L``).
    """

    def __init__(self, variable, weight, aim=None, axes=None, L=None,
                 schedule=None, cross=None, cross_aim=None, cross_L=None):
        self.axes = axes if axes else [""]
        if not isinstance(self.axes, list):
            raise TypeError("The axes must be a list of strings")
        self.axes_len = len(self.axes)

        self.variable = variable
        self.weight = weight
        self.aim = self._as_rows(aim)

        self.schedule = schedule if schedule else range(0)
        self.L = self.arrange_L(L) if L is not None else []

        self.crossed = bool(cross)
        if self.crossed:
            self.cross = cross
            # truthiness test as in the reference (goal.py:78)
            self.cross_aim = self._as_rows(cross_aim if cross_aim else None)
            self.cross_L = self.arrange_L(cross_L) if cross_L is not None else []
        else:
            self.cross = variable
            self.cross_aim = self.aim
            self.cross_L = self.L

    def _as_rows(self, value):
        if value is None:
            value = np.zeros([1, self.axes_len])
        return np.array(value).reshape([-1, self.axes_len])

    def arrange_L(self, L):
        """One matrix per axis; needs an up-to-date schedule (goal.py:88-104)."""
        matrices, problem = per_axis_matrices(L, self.axes_len, self.t if self.schedule else None)
        if problem is not None:
            raise problem
        return matrices

    def update(self, aim=None, weight=None, L=None, schedule=None,
               cross_aim=None, cross_L=None):
        """Replace any of the fields (goal.py:106-136).  The schedule goes first (the L's are
        checked against it); then, in the reference's order, every given field is stored --
        a plain cost keeps its ``cross_*`` twin pointing at the same object, and refuses a
        ``cross_*`` of its own."""
        if schedule is not None:
            self.schedule = schedule
        # field, new value, how it is stored, the twin a plain cost mirrors it into
        steps = (("L", L, self.arrange_L, "cross_L"),
                 ("cross_L", cross_L, self.arrange_L, None),
                 ("aim", aim, self._as_rows, "cross_aim"),
                 ("weight", weight, None, None),
                 ("cross_aim", cross_aim, None, None))
        for field, value, store, twin in steps:
            if value is None:
                continue
            if field.startswith("cross_") and not self.crossed:
                raise KeyError("Trying to set cross_aim in a non-crossed cost")
            kept = store(value) if store else value
            setattr(self, field, kept)
            if twin and not self.crossed:
                setattr(self, twin, kept)

    @property
    def t(self):
        if self.schedule:
            return self.schedule.stop - self.schedule.start
        return None

    def __repr__(self):
        lines = []
        for i, axis in enumerate(self.axes):
            name = self.variable + axis
            for j, goal in enumerate(self.aim[:, i]):
                if j == 0:
                    lines.append(
                        "\n{} --> {}\t\t~ weight = {} ~".format(name, goal, self.weight)
                    )
                else:
                    lines.append(" " * len(name) + " --> " + str(goal))
        return "\n".join(lines) + "\n\n"

    __str__ = __repr__
