"""Recipes for oracle/assemble_port.c  --  TEST INFRASTRUCTURE ONLY.

``Recipe(form, per_instance="LIP")`` walks a formulation exactly as :mod:`oracle.qp_oracle` does
(definition order of ``form.definitions``, ``all_limits``, ``form.goals``; duck-typed on the
reference's attribute names, body.py:149-348) and writes what it finds down for the C interpreter:
which entries of which preview matrix are constants, which are elements of the horizon matrices of
the one dynamics that changes per instance (found by *probing*: the matrices are replaced by index
codes and ``update_definitions()`` shows where each code lands, dynamics.py:277-298), the
combinations of the derived definitions, the limits and the costs with their schedules and ``L``
matrices.  Weights, aims, arrows, centers and extremes go into an array of constants a caller may
override per instance (``const_slice``).

``Recipe.assemble(A, B, given, consts=None)`` runs a batch through the compiled port and returns
``(G, h, P, q)`` stacked over the batch -- the compiled CPU baseline of bench.py and the second
CPU opinion beside the numpy oracle in tests/test_oracle_golden.py.
"""
import ctypes
import os

import numpy as np

from . import qp_oracle as orc

OP_END, OP_BASE, OP_DERIVED, OP_LIMIT, OP_COST = range(5)
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is not built: make -C oracle")
        lib = ctypes.CDLL(path)
        ptr = ctypes.c_void_p
        lib.oracle_assemble_batch.restype = ctypes.c_int
        lib.oracle_assemble_batch.argtypes = [ptr, ptr, ptr, ptr, ptr, ctypes.c_long, ptr, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ptr, ptr, ptr, ptr]
        _lib = lib
    return _lib


def _picked(schedule, rows):
    r = schedule if schedule else range(rows)
    return int(r.start), int(r.step), len(r)


class Recipe:
    def __init__(self, form, per_instance=None):
        maps = orc.qp_index_maps(form.domain, form.optim_variables)
        self.ng, self.no = int(maps["given_len"]), int(maps["optim_len"])
        ng, w = self.ng, self.ng + self.no
        names = list(form.definitions.keys())
        index = {v: i for i, v in enumerate(names)}
        ops, dr, consts = [], [], []
        self._slices = {}

        def pool(values):
            off = len(dr)
            dr.extend(np.asarray(values, dtype=np.float64).ravel().tolist())
            return off

        def const(key, values):
            off = len(consts)
            values = np.asarray(values, dtype=np.float64).ravel()
            consts.extend(values.tolist())
            self._slices[key] = slice(off, off + values.size)
            return off

        # ---- the dynamics whose horizon matrices change per instance: probe where they land -----------
        self.hlen, self.N, self.n, self.m = 0, 0, 0, 0
        probes = None
        if per_instance is not None:
            dyn = form.dynamics[per_instance]
            saved = list(dyn.matrices)
            shapes = [np.shape(M) for M in saved]
            self.m, self.N, self.n = len(saved) - 1, shapes[-1][0], shapes[-1][1]
            if shapes != [(self.N, self.N, self.n)] * self.m + [(self.N, self.n, self.n)]:
                raise NotImplementedError("horizon matrices of %r: %r" % (per_instance, shapes))
            sizes = [int(np.prod(s)) for s in shapes]
            offs = np.concatenate([[0], np.cumsum(sizes)])
            self.hlen = int(offs[-1])
            probes = []
            for shift in (1.0, 1.5):            # (code + 1, code + 1.5: what moves by 0.5 is an element of a matrix)
                dyn.matrices = [(offs[k] + np.arange(sizes[k], dtype=np.float64)).reshape(shapes[k]) + shift
                                for k in range(len(saved))]
                dyn.update_definitions()
                probes.append({v: {dep: np.array(coef, dtype=np.float64) for dep, coef in form.definitions[v].items()}
                               for v in names if v in form.of and form.of[v] == per_instance})
            dyn.matrices = saved
            dyn.update_definitions()

        # ---- a4: preview matrices in definition order (body.py:149-193) ----------------------------------
        rows_of, voff, total = [], [], 0
        for v in names:
            combo = form.definitions[v]
            if v in form.of:
                rows = int(form.dynamics[form.of[v]].all_variables[v])
            else:
                first, coef = next(iter(combo.items()))
                c = np.array(coef)
                rows = rows_of[index[first]] if c.ndim < 2 else c.shape[0]
                if c.ndim == 1:        # (a vector times a matrix: one row, body.py:187-188)
                    rows = 1
            rows_of.append(rows)
            voff.append(total)
            total += rows * w
        for v in names:
            combo, i = form.definitions[v], index[v]
            if v in form.of:
                tmpl_dst, tmpl_val, src = [], [], []
                for dep, coef in combo.items():
                    if dep in maps["given_variables"]:
                        cols = np.arange(ng)[maps["given_ID"][dep]]
                    elif dep in maps["optim_variables"]:
                        cols = ng + np.arange(self.no)[maps["optim_ID"][dep]]
                    else:
                        raise ValueError("The variable {} in the definition of {} seems to not "
                                         "be given nor optimal.".format(dep, v))
                    block = np.zeros((rows_of[i], len(cols))) + np.asarray(coef, dtype=np.float64)   # (numpy's broadcast)
                    dst = (np.arange(rows_of[i])[:, None] * w + cols[None, :]).ravel()
                    moved = np.zeros(block.shape, dtype=bool)
                    if probes is not None and v in probes[0]:
                        p0 = np.zeros(block.shape) + probes[0][v][dep]
                        p1 = np.zeros(block.shape) + probes[1][v][dep]
                        moved = (p1 - p0) == 0.5
                        if np.any((p1 != p0) & ~moved):
                            raise NotImplementedError("definition of %r is no plain view of the horizon matrices" % v)
                        codes = np.rint(p0[moved] - 1.0).astype(np.int64)
                        src.extend(zip(dst[moved.ravel()].tolist(), codes.tolist()))
                    keep = ~moved.ravel()
                    tmpl_dst.extend(dst[keep].tolist())
                    tmpl_val.extend(block.ravel()[keep].tolist())
                ops += [OP_BASE, i, len(tmpl_dst), pool(tmpl_val), len(src)] + tmpl_dst
                for d, s in src:
                    ops += [d, s]
            else:
                terms = list(combo.items())
                ops += [OP_DERIVED, i, len(terms)]
                for dep, coef in terms:
                    c = np.array(coef, dtype=np.float64)
                    if c.ndim == 0:
                        ops += [index[dep], 0, pool(c)]
                    else:
                        c = c.reshape(rows_of[i], rows_of[index[dep]])
                        ops += [index[dep], 1, pool(c)]

        # ---- a7, a8: the limits in the order of the stacked G (body.py:236-264, 304-320) ------------------
        out0, maxrows = 0, 1
        for k, limit in enumerate(orc.all_limits(form)):
            rows = rows_of[index[limit.variable + limit.axes[0]]]
            nl = orc.constraint_nlines(limit)
            nlines = rows if nl is None else int(nl)
            naxes = len(limit.axes)
            start, step, count = _picked(limit.schedule, rows)
            if not limit.L and count != nlines:
                raise NotImplementedError("limit %d: %d lines on %d rows" % (k, nlines, count))
            arrow = np.zeros((nlines, naxes)) + np.asarray(limit.arrow, dtype=np.float64)
            center = np.zeros((nlines, naxes)) + np.asarray(limit.center, dtype=np.float64)
            extreme = np.zeros((nlines, 1)) + np.asarray(limit.extreme, dtype=np.float64)
            ops += [OP_LIMIT, out0, nlines, naxes, const(("limit", k, "arrow"), arrow),
                    const(("limit", k, "center"), center), const(("limit", k, "extreme"), extreme)]
            for a, axis in enumerate(limit.axes):
                loff = pool(np.asarray(limit.L[a], dtype=np.float64).reshape(nlines, count)) if limit.L else -1
                ops += [index[limit.variable + axis], start, step, count, loff]
            out0 += nlines
            maxrows = max(maxrows, nlines, count)
        self.nc = out0

        # ---- a9, a10: the costs (body.py:266-302, 322-329) -------------------------------------------------
        for name, cost in form.goals.items():
            naxes = len(cost.axes)
            rows = rows_of[index[cost.variable + cost.axes[0]]]
            start, step, count = _picked(cost.schedule, rows)
            aim = np.asarray(cost.aim, dtype=np.float64)
            cross_aim = np.asarray(cost.cross_aim, dtype=np.float64)
            if aim.shape != (1, naxes) or cross_aim.shape != (1, naxes):
                raise ValueError("aims are one scalar per axis (body.py:292-300 broadcasts anything else wrongly)")
            ops += [OP_COST, naxes, const(("cost", name, "weight"), cost.weight),
                    const(("cost", name, "aim"), aim), const(("cost", name, "cross_aim"), cross_aim)]
            for a, axis in enumerate(cost.axes):
                lv = pool(np.asarray(cost.L[a], dtype=np.float64)) if cost.L else -1
                lx = pool(np.asarray(cost.cross_L[a], dtype=np.float64)) if cost.cross_L else -1
                if (lv < 0) != (lx < 0):
                    # (an L on one side only: the other side takes the identity)
                    eye = pool(np.eye(count))
                    lv, lx = (eye if lv < 0 else lv), (eye if lx < 0 else lx)
                out_rows = np.asarray(cost.L[a]).shape[0] if cost.L else (
                    np.asarray(cost.cross_L[a]).shape[0] if cost.cross_L else count)
                ops += [index[cost.variable + axis], index[cost.cross + axis], start, step, count, lv, lx, int(out_rows)]
                maxrows = max(maxrows, out_rows, count)
        ops.append(OP_END)

        header = [self.ng, self.no, len(names), total, self.nc, len(consts), self.hlen, self.N, self.n, self.m]
        self.ir = np.asarray(header + rows_of + voff + ops, dtype=np.int32)
        self.dr = np.asarray(dr + [0.0], dtype=np.float64)
        self.consts = np.asarray(consts, dtype=np.float64)
        self.maxrows = int(maxrows)

    def const_slice(self, kind, key, field):
        """Where ``field`` ("weight", "aim", "cross_aim" of a cost by name; "arrow", "center", "extreme" of
        limit number ``key``) sits in the array of constants."""
        return self._slices[(kind, key, field)]

    def assemble(self, A, B, given, consts=None, keep=True):
        """``given``: ``(count, ng)``; ``A``, ``B``: the per-instance system (None without one);
        ``consts``: ``(count, nconst)`` or None (the formulation's own, shared).  Returns
        ``(G, h, P, q)`` of shapes ``(count, nc, no)``, ``(count, nc)``, ``(count, no, no)``,
        ``(count, no)`` (``keep=False``: one slot, overwritten -- timing only)."""
        given = np.ascontiguousarray(given, dtype=np.float64).reshape(-1, self.ng)
        count = given.shape[0]
        if self.hlen:
            A = np.ascontiguousarray(A, dtype=np.float64).reshape(count, self.n, self.n)
            B = np.ascontiguousarray(B, dtype=np.float64).reshape(count, self.n, self.m)
        else:
            A = B = np.zeros(1)
        if consts is None:
            consts, stride = self.consts, 0
        else:
            consts = np.ascontiguousarray(consts, dtype=np.float64).reshape(count, self.consts.size)
            stride = self.consts.size
        slots = count if keep else 1
        Q = np.empty((slots, self.no, self.no))
        q = np.empty((slots, self.no))
        G = np.empty((slots, self.nc, self.no))
        h = np.empty((slots, self.nc))
        rc = load().oracle_assemble_batch(self.ir.ctypes.data, self.dr.ctypes.data, A.ctypes.data, B.ctypes.data,
                                          consts.ctypes.data, stride, given.ctypes.data, count, self.maxrows,
                                          1 if keep else 0, Q.ctypes.data, q.ctypes.data, G.ctypes.data,
                                          h.ctypes.data)
        if rc:
            raise MemoryError("oracle_assemble_batch")
        return G, h, Q, q
