"""QP formulation: the drop-in ``Formulation`` whose numeric work runs on the GPU.

Same public surface as /root/reference/python/mpc_interface/body.py (incorporate_*,
identify_qp_domain, update, make_preview_matrices, arrange_given, preview,
generate_qp_constraint / _cost, generate_all_qp_matrices ...), same return layout
(numpy float64, 2-D column vectors).  The bookkeeping (domain order, sizes, index
ranges) is host Python as in the reference; every matrix is produced by the HIP
kernels of ``mpcasm`` with a batch of one, so this class is the B=1 view of the
batched engine (``mpcasm.engine.Assembler`` is the B>1 view of the same kernels).
There is no numpy fallback: without libmpcasm.so or without a HIP device the
numeric methods raise ``RuntimeError``.
"""
from collections.abc import Mapping

import numpy as np

from . import tools as use


class _PreviewMatrices(Mapping):
    """``Formulation.PM``: ``{definition: (Mg, Mo)}`` computed by the K2 kernel
    on first access and cached until the next ``make_preview_matrices``."""

    def __init__(self, form):
        self._form = form
        self._names = list(form.definitions.keys())
        self._data = None

    def _materialize(self):
        if self._data is None:
            asm = self._form._assembler()
            PM = asm.preview_matrices()[0].cpu().numpy()
            ng = asm.ng
            self._data = {}
            for var in self._names:
                r0, rows = asm.plan.pm_rows[var]
                block = PM[r0:r0 + rows]
                self._data[var] = (np.ascontiguousarray(block[:, :ng]),
                                   np.ascontiguousarray(block[:, ng:]))
        return self._data

    def __getitem__(self, variable):
        return self._materialize()[variable]

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def update(self, other):
        self._materialize().update(other)
        for name in other:
            if name not in self._names:
                self._names.append(name)


class Formulation:
    def __init__(self):
        self.domain = {}            # variable -> size, in incorporation order
        self.optim_variables = []
        self.given_variables = []
        self.optim_sizes = []
        self.given_sizes = []
        self.optim_len = 0          # unknowns of the QP
        self.given_len = 0          # entries of the 'given' vector
        self.optim_ID = {}          # variable -> range in the QP solution
        self.given_ID = {}          # variable -> range in the given vector
        self.domain_ID = {"optim_ID": self.optim_ID, "given_ID": self.given_ID}

        self.definitions = {}       # variable -> LineCombo
        self.dynamics = {}          # name -> ExtendedSystem / DomainVariable
        self.of = {}                # variable -> name of its dynamics

        self.constraint_boxes = {}
        self.constraints = {}       # name -> list of Constraint
        self.goals = {}             # name -> Cost

        self.update_incorporations = use.do_not_update
        self._asm_cache = {}        # structure key -> [Assembler, tick its sources were bound at]
        self._tick = 0              # make_preview_matrices calls so far
        self._frozen = {}           # (dynamics, k) -> horizon matrix as of the last of them
        self._device = None

    # ---- incorporations (body.py:38-94) ------------------------------------
    def incorporate_dynamics(self, name, new_dynamics):
        self.domain.update(new_dynamics.domain)
        self.dynamics[name] = new_dynamics
        for variable in new_dynamics.all_variables.keys():
            self.of[variable] = name
        self.definitions.update(new_dynamics.definitions)

    def incorporate_definition(self, name, new_definition):
        for variable in new_definition.keys():
            assert variable in self.definitions.keys(), (
                "The definition must depend only on defined variables, but the "
                "this definition depends on " + variable
            )
        self.definitions[name] = new_definition

    def incorporate_definitions(self, dict_of_defs):
        for name, combination in dict_of_defs.items():
            self.incorporate_definition(name, combination)

    def _assert_defined(self, variable, axes):
        for axis in axes:
            assert variable + axis in self.definitions.keys(), (
                "All constrained variables must be previously defined, "
                "but '{}' was not defined.".format(variable + axis)
            )

    def incorporate_constraint(self, name, new_limits):
        if not isinstance(new_limits, list):
            new_limits = [new_limits]
        for limit in new_limits:
            self._assert_defined(limit.variable, limit.axes)
        self.constraints[name] = new_limits

    def incorporate_box(self, name, new_box):
        for limit in new_box.constraints:
            self._assert_defined(limit.variable, limit.axes)
        self.constraint_boxes[name] = new_box

    def incorporate_goal(self, name, new_goal):
        self._assert_defined(new_goal.variable, new_goal.axes)
        self.goals[name] = new_goal

    # ---- QP domain bookkeeping (body.py:97-136) -------------------------------
    def identify_qp_domain(self, optimization_domain):
        """``optimization_domain``: names of the unknowns, in QP order; every
        other domain variable is 'given', in domain order."""
        self.optim_variables = optimization_domain
        self.given_variables = [
            v for v in self.domain.keys() if v not in optimization_domain
        ]
        self.update_qp_sizes()
        self.update_qp_IDs()

    def update_qp_domain(self):
        for behavior in self.dynamics.values():
            if behavior.time_variant:
                self.domain.update(behavior.domain)

    def update_qp_sizes(self):
        self.optim_sizes = [self.domain[v] for v in self.optim_variables]
        self.given_sizes = [self.domain[v] for v in self.given_variables]
        self.optim_len = sum(self.optim_sizes)
        self.given_len = sum(self.given_sizes)

    @staticmethod
    def _prefix_ranges(names, sizes):
        ranges, start = {}, 0
        for name, size in zip(names, sizes):
            ranges[name] = range(start, start + size)
            start += size
        return ranges

    def update_qp_IDs(self):
        # dict.update only: keys of variables that left the domain persist, as in
        # the reference (body.py:135-136)
        self.optim_ID.update(self._prefix_ranges(self.optim_variables, self.optim_sizes))
        self.given_ID.update(self._prefix_ranges(self.given_variables, self.given_sizes))

    def set_updating_rule(self, how_to_update=None):
        self.update_incorporations = how_to_update

    def update(self, **kargs):
        """Per-tick refresh (body.py:142-147)."""
        self.update_incorporations(self, **kargs)
        self.update_qp_domain()
        self.update_qp_sizes()
        self.update_qp_IDs()
        self.make_preview_matrices()

    # ---- device plumbing -----------------------------------------------------------
    def _all_limits(self):
        limits = [l for group in self.constraints.values() for l in group]
        limits += [l for box in self.constraint_boxes.values() for l in box.constraints]
        return limits

    _ASM_CACHE_MIN = 24

    def _asm_cache_size(self):
        """Plans kept: room for the whole problem and every single cost / limit (the per-part
        calls of generate_qp_cost / generate_qp_constraint) in three structures -- the walking
        loop alternates between that many (SURVEY: QP 34 / 36 wide) -- and never fewer than
        ``_ASM_CACHE_MIN``."""
        parts = 1 + len(self.goals) + len(self._all_limits())
        return max(self._ASM_CACHE_MIN, 3 * parts)

    def _assembler(self, costs=None, limits=None):
        """Assembler (batch 1) for the whole problem, or for the given costs / limits, carrying
        the numbers of the last ``make_preview_matrices`` (the preview matrices are frozen in
        between, as ``self.PM`` is in the reference) and the current Cost / Constraint
        numbers.  Compiled plans are cached by *structure* (domain, definition coefficients,
        schedules, L contents, field shapes): a tick that keeps the structure -- every tick of
        the walking loop within one phase -- re-uploads the horizon matrices and parameters of a
        cached plan instead of compiling one (the reference re-interprets the description on
        every tick, body.py:142-193; compiling per tick would cost more than that)."""
        from mpcasm.engine import Assembler
        from mpcasm.plan import formulation_key, structure_fingerprint

        whole = costs is None and limits is None
        use_costs = self.goals if whole else ({} if costs is None else costs)
        use_limits = self._all_limits() if whole else ([] if limits is None else limits)
        key = (whole, formulation_key(self), structure_fingerprint(use_costs, use_limits))
        entry = self._asm_cache.get(key)
        if entry is not None:
            self._asm_cache[key] = self._asm_cache.pop(key)          # most recently used: last
            asm, bound_at = entry
            if bound_at != self._tick and not asm.rebind_sources(self, self._frozen):
                entry = None
            elif not asm.refresh_params():
                entry = None
            else:
                entry[1] = self._tick
        if entry is None:
            self._asm_cache.pop(key, None)                            # (a stale entry of this key)
            while len(self._asm_cache) >= self._asm_cache_size():
                self._asm_cache.pop(next(iter(self._asm_cache)))      # least recently used first
            asm = (Assembler(self, batch=1, device=self._device) if whole else
                   Assembler(self, batch=1, device=self._device, costs=use_costs, limits=use_limits))
            asm.rebind_sources(self, self._frozen)
            self._asm_cache[key] = [asm, self._tick]
        return asm

    def make_preview_matrices(self):
        """``self.PM[var] = (Mg, Mo)`` for every definition (body.py:149-193), produced by the
        K2 kernel when first read.  The horizon matrices are frozen here (host copies, a few
        KB): what the kernels read until the next call, whatever the dynamics objects do
        meanwhile -- the reference's ``self.PM`` behaves the same way."""
        self._tick += 1
        self._frozen = {
            (name, k): np.array(M, dtype=np.float64, copy=True)
            for name, dyn in self.dynamics.items()
            for k, M in enumerate(getattr(dyn, "matrices", ()))
        }
        self.PM = _PreviewMatrices(self)

    def get_matrices_from_dynamics(self, variable):
        return self.PM[variable]

    def get_matrices_from_definition(self, variable):
        return self.PM[variable]

    def arrange_given(self, collector):
        """Column vector of all given values, ordered by ``given_ID``
        (body.py:195-207); ``collector`` maps variable -> one-column ndarray."""
        if not self.given_len:
            return np.array([])
        given = np.zeros([self.given_len, 1])
        for variable, indices in self.given_ID.items():
            given[indices] = collector[variable]
        return given

    def _pm_rows(self, variable):
        asm = self._assembler()
        return asm.plan.pm_rows[variable]

    def _preview_rows(self, given, optim):
        """Every row of every definition at ``(given, optim)``: device tensor ``(1, rows)`` from
        ``mpcasm_preview_direct`` (no preview matrix is built), kept until the tick or the point
        changes -- the walking loop reads one variable after the other at the same point."""
        asm = self._assembler()
        g = np.ascontiguousarray(given, dtype=float).reshape(1, -1)
        x = np.ascontiguousarray(optim, dtype=float).reshape(1, -1)
        key = (id(asm), self._tick, g.tobytes(), x.tobytes())
        if getattr(self, "_rows_dev", (None,))[0] != key:
            self._rows_dev = (key, asm.preview_rows(g, x))
        return asm, self._rows_dev[1]

    def preview(self, given, optim, variable, axes=None):
        """``Mg @ given + Mo @ optim`` (body.py:209-219), on the device."""
        asm, rows = self._preview_rows(given, optim)
        values = rows[0].cpu().numpy()

        def rows_of(name):
            r0, rows = asm.plan.pm_rows[name]
            return values[r0:r0 + rows].reshape(-1, 1)

        if axes is None:
            return rows_of(variable)
        return np.hstack([rows_of(variable + axis) for axis in axes])

    def _goal_distances(self, given, optim):
        asm, rows = self._preview_rows(given, optim)
        values = asm.goal_distance(self, rows)[0].cpu().numpy()
        return dict(zip(asm.goal_terms(self)[1], values))

    def goal_distance(self, given, optim, goal_name):
        """Squared distance of a goal's variable to its aim (body.py:221-228): reduced on the
        device (``mpcasm_goal_distance``) from the rows of :meth:`preview`."""
        if goal_name not in self.goals:
            raise KeyError(goal_name)
        return float(self._goal_distances(given, optim)[goal_name])

    def full_goal_distance(self, given, optim):
        """Sum over the goals (body.py:230-234)."""
        dist = self._goal_distances(given, optim)
        value = 0
        for name in self.goals.keys():
            value += float(dist[name])
        return value

    # ---- QP blocks (body.py:236-348) ----------------------------------------------
    @staticmethod
    def _host(t):
        return t[0].cpu().numpy()

    def generate_qp_constraint(self, limit, given):
        """``(A, h)`` of one Constraint; ``limit`` must be up to date."""
        asm = self._assembler(limits=[limit])
        _, _, G, h = asm.assemble(np.asarray(given, dtype=float).reshape(1, -1), want_cost=False)
        return self._host(G), self._host(h).reshape(-1, 1)

    def generate_qp_cost(self, cost, given):
        """``(Q, q)`` of one Cost."""
        asm = self._assembler(costs={"cost": cost})
        P, q, _, _ = asm.assemble(np.asarray(given, dtype=float).reshape(1, -1),
                                  want_constraints=False)
        return self._host(P), self._host(q).reshape(-1, 1)

    def generate_all_qp_constraints(self, given):
        _, _, G, h = self._assembler().assemble(
            np.asarray(given, dtype=float).reshape(1, -1), want_cost=False)
        return self._host(G), self._host(h).reshape(-1, 1)

    def generate_all_qp_costs(self, given):
        P, q, _, _ = self._assembler().assemble(
            np.asarray(given, dtype=float).reshape(1, -1), want_constraints=False)
        return self._host(P), self._host(q).reshape(-1, 1)

    def generate_all_qp_matrices(self, given):
        """``A, h, Q, q`` for ``A x < h`` and ``1/2 x' Q x + q' x`` (body.py:333-348):
        qpsolvers' ``G, h, P, q``; there are no equality constraints."""
        asm = self._assembler()
        # the four results side by side in one device buffer: one copy to the host instead of four
        # (a tick of the walking loop is launch- and copy-bound: ~16 us a copy)
        no, nc = asm.no, asm.nc
        sizes = [no * no, no, nc * no, nc]
        starts = np.cumsum([0] + [n + (n & 1) for n in sizes])            # (16-byte aligned starts)
        flat = getattr(asm, "_flat_out", None)
        if flat is None:
            import torch

            flat = asm._flat_out = torch.empty(int(starts[-1]), dtype=torch.float64, device=asm.device)
        P, q, G, h = (flat[a:a + n].view(shape) for a, n, shape in zip(
            starts, sizes, [(1, no, no), (1, no), (1, nc, no), (1, nc)]))
        asm.assemble(np.asarray(given, dtype=float).reshape(1, -1), out=(P, q, G, h))
        host = flat.cpu().numpy()
        Ph, qh, Gh, hh = (host[a:a + n] for a, n in zip(starts, sizes))
        return Gh.reshape(nc, no), hh.reshape(-1, 1), Ph.reshape(no, no), qh.reshape(-1, 1)
