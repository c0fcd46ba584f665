"""Inequality description: Constraint and Box.

Host-side mirror of /root/reference/python/mpc_interface/restrictions.py.
``Constraint`` holds ``arrow . (V - center) < extreme``; its numeric fields are
per-instance device parameters of the batched assembly and the row-count rule
(:attr:`Constraint.nlines`) is structure.  ``Box`` builds the facets of a convex
polytope with Qhull at set-up time (kept on the host so that the facet order
stays the reference's, SURVEY.md section 8a quirk vii).
"""
import numpy as np

from ._fields import as_rows, per_axis_matrices
import scipy.spatial as sp

from . import tools as use


class Constraint:
    """``arrow * (V - center) < extreme`` with
    ``V = [Lx @ v_x[schedule], Ly @ v_y[schedule], ...]`` (restrictions.py:15-57).

    arrow: ``[m, len(axes)]``; extreme: ``[m, 1]`` or scalar; center:
    ``[m, len(axes)]``; L: list of ``len(axes)`` arrays ``[m, t]``; schedule: a
    ``range`` of ``t`` rows.  Null values: ``L = []``, ``schedule = range(0)``.
    """

    def __init__(self, variable, extreme, axes=None, arrow=None, center=None,
                 L=None, schedule=None):
        self.variable = variable
        self.axes = [""] if axes is None else axes
        if not isinstance(self.axes, list):
            raise TypeError("The axes must be a list of string")
        self.axes_len = len(self.axes)

        self.schedule = range(0) if schedule is None else schedule
        self.L = []
        if L is not None:
            self.arrange_L(L)

        self.extreme = np.array(extreme).reshape([-1, 1]).astype(float)
        self.initialize_geometry(arrow, center)
        self.check_geometry()
        self.normalize()

    # ---- structure -------------------------------------------------------
    def arrange_L(self, L):
        """One matrix per axis; needs an up-to-date schedule (restrictions.py:75-90).  As in
        the reference the new list is stored before it is found wanting."""
        self.L, problem = per_axis_matrices(L, self.axes_len, self.t if self.schedule else None)
        if problem is not None:
            raise problem

    @property
    def m(self):
        return self.L[0].shape[0] if self.L else None

    @property
    def t(self):
        if self.schedule:
            return self.schedule.stop - self.schedule.start
        return None

    def _geometry_rows(self):
        return np.array(
            [self.arrow.shape[0], self.center.shape[0], self.extreme.shape[0]]
        )

    @property
    def nlines(self):
        """Row-count rule (restrictions.py:147-162): rows of L, else length of
        the schedule, else the first non-1 row count among arrow/center/extreme,
        else ``None`` (= as many rows as the variable)."""
        if self.L:
            return self.m
        if self.schedule:
            return self.t
        rows = self._geometry_rows()
        wide = rows[rows != 1]
        return wide[0] if wide.size else None

    # ---- geometry --------------------------------------------------------
    def initialize_geometry(self, arrow, center):
        if arrow is not None:
            self.arrow = np.array(arrow).reshape([-1, self.axes_len])
        elif self.axes_len == 1:
            self.arrow = np.ones(self.extreme.shape)
        else:
            raise ValueError(
                "When using multiple axes, some normal direction 'arrow' "
                "must be provided"
            )

        if center is None:
            self.center = np.zeros([1, self.axes_len])
        else:
            self.center = np.array(center).reshape([-1, self.axes_len])

    def check_geometry(self):
        rows = self._geometry_rows()
        wide = rows[rows != 1]
        if np.any(wide) and np.any(wide != wide[0]):
            raise ValueError(
                "The number of rows in 'arrow', 'center' and 'extreme' must be "
                "equal or 1, but they are {} respectively".format(rows)
            )
        if self.axes_len > 1:
            cols = np.array([self.arrow.shape[1], self.center.shape[1]])
            if np.any(cols != self.axes_len):
                raise IndexError(
                    "'arrow' and 'center' have {} columns but they must have "
                    "{}, one per axis.".format(cols, self.axes_len)
                )

    def normalize(self):
        """Match arrow/extreme row counts and flip signs so that every
        ``extreme >= 0`` (restrictions.py:181-194)."""
        a_rows, e_rows = self.arrow.shape[0], self.extreme.shape[0]
        if a_rows != e_rows:
            if a_rows == 1:
                self.arrow = np.resize(self.arrow, [e_rows, self.axes_len])
            elif e_rows == 1:
                self.extreme = np.resize(self.extreme, [a_rows, 1])

        for i in range(self.extreme.shape[0]):
            if self.extreme[i] < 0:
                self.extreme[i] = -self.extreme[i]
                self.arrow[i] = -self.arrow[i]

    def broadcast(self):
        """Expand 1-row fields to ``nlines`` rows."""
        n = self.nlines
        if n:
            if self.extreme.shape[0] == 1:
                self.extreme = np.resize(self.extreme, [n, 1])
            if self.arrow.shape[0] == 1:
                self.arrow = np.resize(self.arrow, [n, self.axes_len])
            if self.center.shape[0] == 1:
                self.center = np.resize(self.center, [n, self.axes_len])

    # ---- numbers (host view; the kernels fold these in, K4) ---------------
    def matrices(self):
        """Per-axis row coefficients (restrictions.py:175-179)."""
        columns = [self.arrow[:, i][:, None] for i in range(self.axes_len)]
        if self.L:
            return [col * l for col, l in zip(columns, self.L)]
        return columns

    def bound(self):
        """``extreme + sum_axes arrow * center`` (restrictions.py:198-199)."""
        return self.extreme + (self.arrow * self.center).sum(axis=1).reshape([-1, 1])

    def update(self, extreme=None, arrow=None, center=None, L=None, schedule=None):
        """Replace any of the fields (restrictions.py:201-219): schedule first (L is checked
        against it), then the numbers; new geometry is checked, and normalised when a sign may
        have changed (new ``extreme`` or ``arrow``)."""
        if schedule is not None:
            self.schedule = schedule
        if L is not None:
            self.arrange_L(L)
        numbers = {"extreme": (extreme, 1), "arrow": (arrow, self.axes_len),
                   "center": (center, self.axes_len)}
        changed = [name for name, (value, _) in numbers.items() if value is not None]
        for name in changed:
            value, columns = numbers[name]
            setattr(self, name, as_rows(value, columns))
        if changed:
            self.check_geometry()
            if "extreme" in changed or "arrow" in changed:
                self.normalize()

    # ---- feasibility checks (restrictions.py:221-257) ----------------------
    def is_feasible(self, points, space="SS"):
        check = {"SS": self._is_feasible_SS, "TS": self._is_feasible_TS}.get(space)
        if check is None:
            return None
        if isinstance(points, list):
            return [check(point) for point in points]
        return check(points)

    def SS_to_TS(self, ss_point):
        """``ss_point`` has one column per task-space axis."""
        if self.L:
            return np.vstack(
                [l @ p.T for l, p in zip(self.L, np.transpose(ss_point))]
            ).T
        return ss_point

    def _is_feasible_SS(self, ss_point):
        return self._is_feasible_TS(self.SS_to_TS(ss_point))

    def _is_feasible_TS(self, ts_point):
        return np.sum(self.arrow * (ts_point - self.center), axis=1) < self.extreme

    def __repr__(self):
        axes = "" if self.axes == [""] else "_" + "".join(a[1:] for a in self.axes)
        text = "\nvariable: " + self.variable + axes
        if self.L != []:
            text += "\nwith L = " + ",\n".join(str(l) for l in self.L)
        for label, pad, field in (("arrow", 7, self.arrow),
                                  ("center", 8, self.center),
                                  ("extreme", 9, self.extreme)):
            text += "\n\t\t\t\t\t{}: ".format(label) + (" " * pad).join(
                str(row) for row in field
            )
        return text + "\n"

    __str__ = __repr__


def _check_ss_shape(box, value):
    allowed = ((box.ss_dimention, box.ts_dimention), (box.ts_dimention,))
    if np.shape(value) not in allowed:
        raise ValueError(
            "The 'new_center' must have {} rows and {} columns, but its shape "
            "is {}".format(box.ss_dimention, box.ts_dimention, np.shape(value))
        )


class Box:
    """A set of facet :class:`Constraint`s sharing a variable (restrictions.py:287-489)."""

    def __init__(self, time_variant=None, how_to_update=None):
        self.constraints = []

        self.ts_vertices = np.array([])
        self.ss_vertices = np.array([])
        self.ts_center = np.array([])
        self.ts_orientation = []
        self.ss_center = np.array([])
        self.ss_orientation = []

        self.scale_factor = np.array([1.0])
        self.schedule = range(0)
        self.safety_margin = 0
        self.axes = [""]
        self.ss_dimention = 0
        self.ts_dimention = 0

        self.time_variant = time_variant
        self._updater = (
            how_to_update
            if (how_to_update is not None and time_variant)
            else use.do_not_update
        )

    @classmethod
    def task_space(cls, variable, vertices, axes=None, L=None, schedule=None,
                   time_variant=None, how_to_update=None):
        """One facet constraint per hull simplex of ``vertices`` (restrictions.py:313-340)."""
        box = cls(time_variant, how_to_update)
        arrows, extremes, center = box_boundaries(vertices)
        for arrow, extreme in zip(arrows, extremes):
            box.constraints.append(
                Constraint(variable, extreme, axes, arrow, center, L, schedule)
            )
        dim = vertices.shape[1]
        box.ss_dimention = box.ts_dimention = dim
        box.axes = box.constraints[0].axes
        box.ts_center = center
        box.ss_center = np.zeros(center.shape)
        box.ts_orientation = [np.eye(dim)]
        box.ss_orientation = [np.eye(dim)]
        box.ts_vertices = vertices
        box.schedule = schedule
        return box

    @classmethod
    def state_space(cls, variable, vertices, axes=None, schedule=None,
                    time_variant=None, how_to_update=None):
        """Polytope on the rows of one variable: the facet normals become
        ``L`` (restrictions.py:342-378)."""
        box = cls(time_variant, how_to_update)
        arrows, extremes, center_ss = box_boundaries(vertices)
        offsets = np.sum(arrows * center_ss, axis=1).reshape([-1, 1])

        for arrow, extreme, offset in zip(arrows, extremes, offsets):
            box.constraints.append(
                Constraint(variable, extreme, axes=axes, center=offset, L=arrow,
                           schedule=schedule)
            )
        box.ss_dimention = vertices.shape[1]
        box.ts_dimention = 1
        box.axes = box.constraints[0].axes
        box.ss_center = center_ss.reshape([-1, 1])
        box.ts_center = 0
        box.ss_orientation = [np.eye(box.ss_dimention)]
        box.ts_orientation = [1]
        box.ss_vertices = vertices
        box.schedule = schedule
        return box

    # ---- transforms (restrictions.py:380-486) -----------------------------
    def recenter_in_TS(self, new_center):
        """Beware: for a state-space box this deforms the set; prefer
        :meth:`recenter_in_SS`."""
        self.ts_center = np.array(new_center)
        for facet in self.constraints:
            facet.update(center=self.ts_center)

    def recenter_in_SS(self, new_center):
        _check_ss_shape(self, new_center)
        self.ss_center = np.array(new_center)
        for facet in self.constraints:
            facet.update(center=facet.SS_to_TS(new_center))

    def reschedule(self, new_schedule):
        self.schedule = new_schedule
        for facet in self.constraints:
            facet.update(schedule=new_schedule)

    def translate_in_TS(self, translation):
        self.ts_center += translation
        for facet in self.constraints:
            facet.update(center=facet.center + translation)

    def translate_in_SS(self, translation):
        _check_ss_shape(self, translation)
        self.ss_center += np.array(translation)
        for facet in self.constraints:
            facet.update(center=facet.center + facet.SS_to_TS(translation))

    def rotate_in_TS(self, rotations):
        nlines = self.constraints[0].nlines
        count = 1 if nlines is None else nlines

        if not isinstance(rotations, list):
            rotations = [rotations] * count
        if len(rotations) not in (count, 1):
            raise IndexError(
                "'rotations' must contain 1 or {} rotation matrices".format(count)
            )
        if len(rotations) == 1:
            rotations = rotations * count

        for facet in self.constraints:
            arrows = facet.arrow
            if arrows.shape[0] == 1 and count > 1:
                arrows = np.resize(arrows, (count, facet.axes_len))
            facet.update(
                arrow=np.vstack([a @ R.T for R, a in zip(rotations, arrows)])
            )

    def rotate_in_SS(self, rotations):
        raise NotImplementedError("Maybe later.")

    def is_feasible(self, points, space="SS"):
        if not isinstance(points, list):
            points = [points]
        return [
            all(facet.is_feasible(point, space) for facet in self.constraints)
            for point in points
        ]

    def scale_box(self, scale_factor):
        for facet in self.constraints:
            facet.update(extreme=facet.extreme * scale_factor / self.scale_factor)
        self.scale_factor = scale_factor

    def set_safety_margin(self, margin):
        for facet in self.constraints:
            facet.update(extreme=facet.extreme - margin * np.linalg.norm(facet.arrow))
        self.safety_margin = margin

    def update(self, **kargs):
        self._updater(self, **kargs)


def box_boundaries(vertices):
    """Outward unit normals, offsets and centroid of the convex hull of
    ``vertices`` (one vertex per row), restrictions.py:492-531.  The facet
    order is Qhull's simplex order."""
    vertices = vertices.astype("float64")
    count, dim = vertices.shape
    center = vertices.sum(axis=0) / count

    if dim == 1:
        simplices = np.array([[0], [1]])
        arrows = np.ones([2, 1])
    else:
        simplices = sp.ConvexHull(vertices).simplices
        if dim == 2:
            edge = vertices[simplices[:, 0]] - vertices[simplices[:, 1]]
            arrows = np.column_stack([edge[:, 1], -edge[:, 0]])
        elif dim == 3:
            e0 = vertices[simplices[:, 0]] - vertices[simplices[:, 2]]
            e1 = vertices[simplices[:, 1]] - vertices[simplices[:, 2]]
            arrows = np.column_stack([
                e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1],
                e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2],
                e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0],
            ])

    for i in range(arrows.shape[0]):
        arrows[i] = arrows[i] / np.linalg.norm(arrows[i])

    on_facet = vertices[simplices[:, 0]]
    extremes = np.sum(arrows * (on_facet - center), axis=1).reshape([-1, 1])
    return arrows, extremes, center
