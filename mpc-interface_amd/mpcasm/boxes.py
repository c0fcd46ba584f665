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

    def scale_box(self, scale_factor):
        """restrictions.py:474-478: extremes times ``scale_factor / previous factor``."""
        torch = self.asm._torch
        new = torch.as_tensor(scale_factor, dtype=torch.float64, device=self.asm.device).reshape(-1)
        new = new.expand(self.asm.batch).contiguous()
        self._run(capi.BOX_SCALE, new / self.scale_factor, 1)
        self.scale_factor = new.clone()

    def set_safety_margin(self, margin):
        """restrictions.py:480-486: extremes shrink by ``margin * norm(arrow)`` (the
        Frobenius norm of a facet's whole arrow field, as ``np.linalg.norm`` gives it)."""
        self._run(capi.BOX_MARGIN, margin, 1)
