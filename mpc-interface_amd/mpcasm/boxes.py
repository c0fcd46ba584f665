"""Batched box transforms.

The reference moves, turns, scales and shrinks a ``Box`` between ticks by rewriting the
``arrow`` / ``center`` / ``extreme`` fields of its facets, one walker at a time
(restrictions.py:380-486, each ending in ``Constraint.update`` -> ``normalize``,
restrictions.py:180-219).  Here the same updates run on the device for a whole batch: the
facets' numbers are per-instance parameters of an :class:`~mpcasm.engine.Assembler`
(``asm.params``), and one launch of ``mpcasm_box_transform`` rewrites them in place.
"""
import ctypes

import numpy as np

from . import capi


class BoxBatch:
    """The facets of ``form.constraint_boxes[name]`` inside ``asm.params``.

    Every method takes one argument per instance (leading dimension ``asm.batch``) or one
    for all.  Shapes of the facets' fields are those the plan was compiled with: a rotation
    or a margin that would need one arrow per row of a single-row arrow field (the reference
    resizes it, restrictions.py:181-188) asks for a formulation whose facets already carry
    per-row arrows.
    """

    def __init__(self, asm, form, name):
        self.asm = asm
        torch = asm._torch
        first = sum(len(group) for group in form.constraints.values())
        for other, box in form.constraint_boxes.items():
            if other == name:
                break
            first += len(box.constraints)
        else:
            raise KeyError(name)
        box = form.constraint_boxes[name]
        records = []
        for k in range(first, first + len(box.constraints)):
            a0, arows, axes = asm.plan.param_slots[("limit", k, "arrow")]
            c0, crows, _ = asm.plan.param_slots[("limit", k, "center")]
            e0, erows, _ = asm.plan.param_slots[("limit", k, "extreme")]
            if arows != erows and arows != 1:
                raise ValueError("facet %d: %d arrows for %d extremes" % (k, arows, erows))
            records.append([a0, arows, c0, crows, e0, erows, axes])
        self.axes = records[0][6]
        if self.axes > 4:
            raise ValueError("boxes of more than 4 axes are not supported")
        self._records = records
        # state-space boxes (Box.state_space, or a task-space box built with L): the facets' L,
        # [nfacets][axes][lrows][ss_dim] -- part of the structure, the same for every instance
        self._L = None
        Ls = [facet.L for facet in box.constraints]
        if all(Ls):
            stack = np.asarray([[np.atleast_2d(np.asarray(l, dtype=np.float64)) for l in L] for L in Ls])
            if stack.ndim == 4 and stack.shape[1] == self.axes:
                self._L = torch.as_tensor(np.ascontiguousarray(stack), device=asm.device)
                self.lrows, self.ss_dim = int(stack.shape[2]), int(stack.shape[3])
        self._facets = torch.as_tensor(np.asarray(records, dtype=np.int32), device=asm.device)
        self.nfacets = len(records)
        self.scale_factor = torch.ones(asm.batch, dtype=torch.float64, device=asm.device)

    def _run(self, op, arg, width):
        asm, torch = self.asm, self.asm._torch
        arg = torch.as_tensor(arg, dtype=torch.float64, device=asm.device).reshape(-1, width)
        if arg.shape[0] not in (1, asm.batch):
            raise ValueError("expected 1 or %d instances, got %d" % (asm.batch, arg.shape[0]))
        arg = arg.contiguous()
        with torch.cuda.device(asm.device):
            rc = capi.load().mpcasm_box_transform(
                asm.params.data_ptr(), asm.params.shape[1], asm.batch, self._facets.data_ptr(),
                self.nfacets, op, arg.data_ptr(), width if arg.shape[0] == asm.batch else 0,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        capi.check(rc, "mpcasm_box_transform")

    def recenter_in_TS(self, new_center):
        """restrictions.py:380-388: every facet's centre becomes ``new_center`` ``(B, axes)``."""
        self._run(capi.BOX_RECENTER, new_center, self.axes)

    def translate_in_TS(self, translation):
        """restrictions.py:411-415: centres move by ``translation`` ``(B, axes)``."""
        self._run(capi.BOX_TRANSLATE, translation, self.axes)

    def rotate_in_TS(self, rotations):
        """restrictions.py:436-455 with one rotation matrix per instance ``(B, axes, axes)``,
        applied to every row: ``arrow_row <- arrow_row @ R.T``."""
        self._run(capi.BOX_ROTATE, rotations, self.axes * self.axes)

    def _run_ss(self, op, point):
        asm, torch = self.asm, self.asm._torch
        if self._L is None:
            raise ValueError("this box has no L matrices: it lives in the task space only")
        if any(rec[3] != self.lrows for rec in self._records):
            raise ValueError("the facets' centre fields were compiled with %s rows, a state-space "
                             "point gives %d" % (sorted({rec[3] for rec in self._records}), self.lrows))
        width = self.ss_dim * self.axes
        point = torch.as_tensor(point, dtype=torch.float64, device=asm.device)
        if point.shape[-2:] != (self.ss_dim, self.axes) and not (
                self.axes == 1 and point.shape[-1] == self.ss_dim):
            # the reference's message (restrictions.py:393-400)
            raise ValueError("The 'new_center' must have {} rows and {} columns, but its shape "
                             "is {}".format(self.ss_dim, self.axes, tuple(point.shape)))
        point = point.reshape(-1, width)
        if point.shape[0] not in (1, asm.batch):
            raise ValueError("expected 1 or %d instances, got %d" % (asm.batch, point.shape[0]))
        point = point.contiguous()
        with torch.cuda.device(asm.device):
            rc = capi.load().mpcasm_box_transform_ss(
                asm.params.data_ptr(), asm.params.shape[1], asm.batch, self._facets.data_ptr(),
                self.nfacets, op, self._L.data_ptr(), self.lrows, self.ss_dim, point.data_ptr(),
                width if point.shape[0] == asm.batch else 0,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        capi.check(rc, "mpcasm_box_transform_ss")

    def recenter_in_SS(self, new_center):
        """restrictions.py:390-404: every facet's centre becomes ``SS_to_TS(new_center)``,
        ``new_center`` ``(B, ss_dimention, ts_dimention)`` -- one column per task-space axis."""
        self._run_ss(capi.BOX_RECENTER, new_center)

    def translate_in_SS(self, translation):
        """restrictions.py:417-431: centres move by ``SS_to_TS(translation)``."""
        self._run_ss(capi.BOX_TRANSLATE, translation)

    def _per_row_arrows(self, what):
        # Constraint.normalize gives every extreme row its own arrow row before it flips signs
        # (restrictions.py:181-194); a facet compiled with ONE arrow for several extremes has
        # no room for that in the parameters
        if any(rec[1] == 1 and rec[5] > 1 for rec in self._records):
            raise ValueError("%s may flip single rows of a facet: build the box with one arrow "
                             "per row (arrow of shape (rows, axes))" % what)

    def scale_box(self, scale_factor):
        """restrictions.py:474-478: extremes times ``scale_factor / previous factor``."""
        self._per_row_arrows("scale_box")
        torch = self.asm._torch
        new = torch.as_tensor(scale_factor, dtype=torch.float64, device=self.asm.device).reshape(-1)
        new = new.expand(self.asm.batch).contiguous()
        self._run(capi.BOX_SCALE, new / self.scale_factor, 1)
        self.scale_factor = new.clone()

    def set_safety_margin(self, margin):
        """restrictions.py:480-486: extremes shrink by ``margin * norm(arrow)`` (the
        Frobenius norm of a facet's whole arrow field, as ``np.linalg.norm`` gives it)."""
        self._per_row_arrows("set_safety_margin")
        self._run(capi.BOX_MARGIN, margin, 1)
