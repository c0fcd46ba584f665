"""Batched tick driver for a fleet of biped walkers.

The reference's walking loop (use_examples/simple_functional_example/
biped_mpc_loop.py:17-95) advances ONE walker: every tick it counts the step times
down, re-plans the step indicator matrix (tools.update_step_matrices -> plan_steps),
moves the stepping-area centres (tools.update_stepping_area) and re-assembles the QP,
whose width changes with the walking phase (34 / 36 unknowns at N=16).

Here a whole fleet advances in lock-step ticks, every walker with its own phase.
The QP *structure* depends only on the number ``p`` of steps inside the preview, so
the fleet is split into structure buckets (one compiled plan and one persistent
assembly launch per bucket and tick); what differs between walkers of a bucket is
numbers: the step indicator matrix ``E`` (a per-instance source), the stepping-area
centres (per-instance parameters) and the given vector.
"""
import numpy as np

from . import problems


def steps_in_preview(step_times, N):
    """Boolean mask of the step instants inside ``[0, N - 1)`` (tools.py:84 with count=0)."""
    return (step_times >= 0) & (step_times < N - 1)


def step_indicator(step_times, N):
    """``E[b, k, s] = 1`` when preview sample ``k`` lies after the ``s``-th step instant of
    walker ``b`` (tools.plan_steps, tools.py:79-101, for walkers that all have the same
    number of steps in the preview).  ``step_times``: ``(B, p)`` kept instants."""
    k = np.arange(N).reshape(1, N, 1)
    return (k > step_times[:, None, :]).astype(np.float64)


def stepping_centers(step_count, p, xy):
    """Alternating left/right centres of the next ``p`` stepping areas for every walker
    (tools.find_step_centers, tools.py:158-165): ``(B, p, 2)``."""
    side = (-1.0) ** (np.asarray(step_count) + 1)
    alt = np.tile([1.0, -1.0], p // 2 + 1)[:p]
    out = np.empty((len(side), p, 2))
    out[:, :, 0] = xy[0]
    out[:, :, 1] = side[:, None] * alt[None, :] * xy[1]
    return out


class FleetClock:
    """Vectorised step-time bookkeeping (biped_mpc_loop.py:41-45): every tick the step
    times count down; when the first reaches -1 they wrap by ``step_samples`` and the
    walker's step count goes up."""

    def __init__(self, step_samples, n_steps, phases):
        phases = np.asarray(phases, dtype=np.int64)
        base = np.array([(i + 1) * step_samples - 1 for i in range(n_steps)], dtype=np.int64)
        self.n = int(step_samples)
        self.step_times = base[None, :] - phases[:, None]
        self.step_count = np.zeros(len(phases), dtype=np.int64)

    def tick(self):
        self.step_times -= 1
        wrap = self.step_times[:, 0] == -1
        self.step_times[wrap] += self.n
        self.step_count[wrap] += 1


class WalkerFleet:
    """``batch`` walkers on the biped formulation of ``problems.biped``.

    ``phases[b]`` in ``[0, step_samples)`` is how many ticks walker ``b`` is ahead in its
    step cycle.  :meth:`tick` assembles the QPs of all walkers for the current tick and
    advances the clocks; it returns one entry per structure bucket:
    ``{"p": steps in preview, "index": walker ids, "P", "q", "G", "h": device tensors}``.
    """

    def __init__(self, batch, phases=None, conf=None, api=None, device=None, graphs=False, side_by_side=False):
        from .engine import Assembler, require_device

        self._torch = require_device()
        self.conf = conf or problems.BipedConfig()
        self.api = api or problems.load_api("mpc_interface")
        self.batch = int(batch)
        n = self.conf.step_samples
        self.N = self.conf.horizon_lenght
        phases = np.arange(self.batch) % n if phases is None else np.asarray(phases)
        self.clock = FleetClock(n, self.conf.num_steps, phases)
        self.device = device
        self._ticks, self._cache = 0, {}
        # graphs=True: a tick's launches (parameter update, gather of `given`, one assembly per
        # structure bucket) are captured in a hipGraph the first time its place in the step cycle
        # comes round and replayed from then on -- the tick is a copy of `given` into a fixed
        # buffer and one graph launch instead of a dozen host-side calls
        self._use_graphs, self._graphs, self._given = bool(graphs), {}, None
        # side_by_side=True: inside the graph every bucket on a branch of its own with its share of the chip's
        # workgroup slots (capi.OPT_RESIDENT_GRID, by the number of its walkers).  Measured (round 4, 4 096
        # walkers: buckets of 512 and 3 584): 64 us per tick against 48 one after the other -- this runtime
        # replays a graph with parallel branches through signals between queues that cost more than the
        # second set-up they would hide; off by default
        self._side, self._side_by_side = [], bool(side_by_side)

        # one template formulation + assembler per structure bucket (steps in preview)
        self.buckets = {}
        for phi in range(n):
            times = np.array([(i + 1) * n - 1 - phi for i in range(self.conf.num_steps)])
            p = int(steps_in_preview(times, self.N).sum())
            if p in self.buckets:
                continue
            form = problems.biped(self.api, self.conf)
            form.update(step_times=times, step_count=0)
            asm = Assembler(form, batch=self.batch, device=device)
            torch = self._torch
            E = torch.zeros((self.batch, self.N, p, 1), dtype=torch.float64, device=asm.device)
            asm.bind_source(("steps", 0), E)
            # the stepping area is the first box: its facets are the first limits
            n_facets = len(form.constraint_boxes["stepping area"].constraints)
            first = sum(len(group) for group in form.constraints.values())
            facets = range(first, first + n_facets)
            # parameter columns of the centres of all facets of the stepping area, side by side
            cols = []
            for k in facets:
                sl, _ = asm.param_slice("limit", k, "center")
                cols.extend(range(sl.start, sl.stop))
            self.buckets[p] = dict(form=form, asm=asm, E=E, facets=facets,
                                   center_cols=torch.as_tensor(cols, dtype=torch.int64, device=asm.device))

    @property
    def given_len(self):
        return next(iter(self.buckets.values()))["asm"].ng

    def structure_of(self):
        """Steps in the preview of every walker right now (its structure bucket)."""
        return steps_in_preview(self.clock.step_times, self.N).sum(axis=1)

    def _bucket_inputs(self):
        """This tick's per-bucket inputs on the device: walker ids, step indicator matrices,
        stepping-area centres.  The fleet advances in lock step, so they repeat with period
        ``2 * step_samples`` (the step cycle times the left/right alternation): each of those
        ticks is worked out once on the host (tools.plan_steps / find_step_centers semantics)
        and kept on the device in the shape the assembler takes as it is -- the indicator
        matrices as a full source tensor (bound per tick, no copy), the parameters with this place's
        centres of all facets in their columns (handed over per tick, no copy), the walkers' ids as the
        index of their rows of ``given``."""
        key = self._ticks % (2 * self.conf.step_samples)
        if key not in self._cache:
            torch = self._torch
            p_of = self.structure_of()
            entry = []
            for p, bucket in self.buckets.items():
                idx = np.nonzero(p_of == p)[0]
                if idx.size == 0:
                    continue
                asm = bucket["asm"]
                dev = asm.device
                times = self.clock.step_times[idx]
                kept = times[steps_in_preview(times, self.N)].reshape(idx.size, p)
                E = torch.zeros((self.batch, self.N, p, 1), dtype=torch.float64, device=dev)
                E[:idx.size, :, :, 0] = torch.as_tensor(step_indicator(kept, self.N), device=dev)
                centers = stepping_centers(self.clock.step_count[idx], p, self.conf.stepping_center)
                centers = torch.as_tensor(centers.reshape(idx.size, -1), device=dev)
                # the bucket's parameters at this place of the cycle: the assembler's own, with the centres
                # of the stepping area of these walkers in their columns -- a tensor per place, so that a
                # tick copies nothing (only the centres change with the place in the cycle)
                params = asm.params.clone()
                params[:idx.size].index_copy_(1, bucket["center_cols"], centers.repeat(1, len(bucket["facets"])))
                entry.append(dict(p=p, idx=idx, index=torch.as_tensor(idx, dtype=torch.int32, device=dev), E=E,
                                  params=params))
            self._cache[key] = entry
        return self._cache[key]

    def _launch(self, given, side_by_side=False):
        """This tick's launches for ``given`` (a device tensor): ONE assembly per structure bucket
        (its walkers' rows of ``given`` picked by index inside the kernel, the parameters of this place in
        the step cycle kept ready).  ``side_by_side``: every bucket on a stream of its own with its share
        of the workgroup slots (forked from and joined to the current stream: what a graph captures as
        parallel branches)."""
        from . import capi
        torch = self._torch
        items = self._bucket_inputs()
        side_by_side = side_by_side and len(items) > 1
        out = []
        if side_by_side:
            dev = self.buckets[items[0]["p"]]["asm"].device
            cur = torch.cuda.current_stream(dev)
            while len(self._side) < len(items) - 1:
                self._side.append(torch.cuda.Stream(device=dev))
            total = sum(item["idx"].size for item in items)
            slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
        for i, item in enumerate(items):
            p, idx = item["p"], item["idx"]
            bucket = self.buckets[p]
            asm = bucket["asm"]
            asm.bind_source(("steps", 0), item["E"])
            stream = None
            if side_by_side:
                asm.set_option(capi.OPT_RESIDENT_GRID, max(1, int(round(slots * idx.size / total))))
                if i:
                    stream = self._side[i - 1]
                    stream.wait_stream(cur)
            elif self._side:
                asm.set_option(capi.OPT_RESIDENT_GRID, -1)
            # (no gather, no copy: the walkers' rows of `given` by index, the place's own parameters)
            P, q, G, h = asm.assemble(given, count=idx.size, index=item["index"], params=item["params"],
                                      stream=stream)
            out.append({"p": p, "index": idx, "P": P[:idx.size], "q": q[:idx.size],
                        "G": G[:idx.size], "h": h[:idx.size]})
        if side_by_side:
            for stream in self._side[:len(items) - 1]:
                cur.wait_stream(stream)
        return out

    def given_buffer(self):
        """The fleet's own ``(batch, ng)`` buffer of ``given`` on the device: a caller that writes the
        walkers' states straight into it and hands it to :meth:`tick` saves the copy a replayed graph needs
        (its launches read fixed addresses)."""
        if self._given is None:
            dev = next(iter(self.buckets.values()))["asm"].device
            self._given = self._torch.empty((self.batch, self.given_len), dtype=self._torch.float64, device=dev)
        return self._given

    def tick(self, given):
        """Assemble this tick's QPs (``given``: ``(batch, ng)`` tensor or array), then
        advance every walker's clock.  The results live in the assemblers' own buffers: they
        are valid until the next tick."""
        torch = self._torch
        dev = next(iter(self.buckets.values()))["asm"].device
        g = given if isinstance(given, torch.Tensor) else torch.as_tensor(
            np.asarray(given, dtype=np.float64), device=dev)
        g = g.to(dev)
        if not self._use_graphs:
            out = self._launch(g)
        else:
            if g is not self.given_buffer() and g.data_ptr() != self._given.data_ptr():
                self._given.copy_(g, non_blocking=True)
            key = self._ticks % (2 * self.conf.step_samples)
            if key not in self._graphs:
                self._launch(self._given)            # once as it is: kernels compiled, buffers there
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._launch(self._given, side_by_side=self._side_by_side)
                self._graphs[key] = (graph, out)
            graph, out = self._graphs[key]
            graph.replay()
        self.clock.tick()
        self._ticks += 1
        return out
