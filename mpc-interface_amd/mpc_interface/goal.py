"""Cost description.

Host-side mirror of /root/reference/python/mpc_interface/goal.py:11-161.  A
``Cost`` is a pure data holder; its numeric fields (weight, aim, cross_aim)
become per-instance device parameters of the batched assembly
(``mpcasm.plan``), its ``L``/``schedule`` become row-set structure.
"""
import numpy as np


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
        per_axis = L if isinstance(L, list) else [L]
        if len(per_axis) == 1:
            per_axis = per_axis * self.axes_len
        elif len(per_axis) not in (self.axes_len, 0):
            raise IndexError(
                "'L' must have 0, 1 or len(axes) = {} elements".format(self.axes_len)
            )
        if self.schedule and np.any([l.shape[-1] != self.t for l in per_axis]):
            raise ValueError(
                "arrays in L must have {} columns, which is given by the "
                "'schedule'.".format(self.t)
            )
        return per_axis

    def update(self, aim=None, weight=None, L=None, schedule=None,
               cross_aim=None, cross_L=None):
        """goal.py:106-136 (same ordering: schedule, L, cross_L, aim, weight,
        cross_aim)."""
        if schedule is not None:
            self.schedule = schedule

        if L is not None:
            self.L = self.arrange_L(L)
            if not self.crossed:
                self.cross_L = self.L

        if cross_L is not None:
            if not self.crossed:
                raise KeyError("Trying to set cross_aim in a non-crossed cost")
            self.cross_L = self.arrange_L(cross_L)

        if aim is not None:
            self.aim = np.array(aim).reshape([-1, self.axes_len])
            if not self.crossed:
                self.cross_aim = self.aim

        if weight is not None:
            self.weight = weight

        if cross_aim is not None:
            if not self.crossed:
                raise KeyError("Trying to set cross_aim in a non-crossed cost")
            self.cross_aim = cross_aim

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
