"""Batched QP assembly on one MI355X: torch-ROCm tensors as device buffers, the
kernels of libmpcasm.so through the C ABI (``mpcasm.capi``).

* :func:`fill_su` -- K1, batched ``tools.extend_matrices`` (tools.py:14-33)
* :class:`Assembler` -- K2+K3+K4 for a batch of instances of one Formulation
  structure: ``assemble(given)`` returns the qpsolvers blocks
  ``P (B,no,no), q (B,no), G (B,nc,no), h (B,nc)`` (body.py:333-348) and
  ``preview_matrices()`` the stacked ``[Mg | Mo]`` of every definition
  (body.py:149-193).

torch is used for memory, streams and ``data_ptr()`` only; nothing here
computes on the host and nothing falls back to the CPU.
"""
import ctypes

import numpy as np

from . import capi
from .plan import compile_plan, is_causal


def _torch():
    import torch

    return torch


def require_device():
    """The torch module, after checking that a HIP device is usable."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "mpcasm: no HIP device is visible; the QP-assembly path runs only on the GPU "
            "(there is no CPU fallback)")
    capi.load()
    return torch


def _stream_handle(torch, stream):
    if stream is None:      # (the raw handle of the current stream: no Stream object per launch)
        try:
            return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        except AttributeError:
            stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream)


def _as_device(torch, x, device):
    """float64 contiguous device tensor from a tensor or array."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=device)


# --------------------------------------------------------------------------
# K1
# --------------------------------------------------------------------------
def fill_su(A, B, N, ltv=False, out=None, stream=None, device=None):
    """Horizon matrices of a batch of systems (``mpcasm_fill_su``).

    ``A``: ``(B, n, n)`` and ``B``: ``(B, n, m)`` (or ``(B, N, n, n)`` /
    ``(B, N, n, m)`` when ``ltv``).  Returns device tensors ``S (B, N, n, n)``
    and ``U (B, m, N, N, n)`` with ``U[b, j]`` = the reference's ``U[j]``.
    """
    torch = require_device()
    if device is None:
        device = A.device if isinstance(A, torch.Tensor) and A.is_cuda else torch.device(
            "cuda", torch.cuda.current_device())
    A = _as_device(torch, A, device)
    Bm = _as_device(torch, B, device)
    N = int(N)
    if ltv:
        if A.dim() != 4 or Bm.dim() != 4 or A.shape[1] != N or Bm.shape[1] != N:
            raise ValueError("ltv fill needs A (B,N,n,n) and B (B,N,n,m)")
        batch, _, n, m = Bm.shape
    else:
        if A.dim() != 3 or Bm.dim() != 3:
            raise ValueError("fill needs A (B,n,n) and B (B,n,m)")
        batch, n, m = Bm.shape
    if A.shape[-2:] != (n, n) or A.shape[0] != batch:
        raise ValueError("A %s does not match B %s" % (tuple(A.shape), tuple(Bm.shape)))
    if out is None:
        S = torch.empty((batch, N, n, n), dtype=torch.float64, device=device)
        U = torch.empty((batch, m, N, N, n), dtype=torch.float64, device=device)
    else:
        S, U = out
    with torch.cuda.device(device):
        rc = capi.load().mpcasm_fill_su(
            A.data_ptr(), Bm.data_ptr(), S.data_ptr(), U.data_ptr(), batch, N, n, m,
            1 if ltv else 0, _stream_handle(torch, stream))
    capi.check(rc, "mpcasm_fill_su")
    return S, U


def fill_su_numpy(A, B, N, ltv=False):
    """:func:`fill_su` with numpy in / numpy out (single-instance drop-in path)."""
    S, U = fill_su(A, B, N, ltv=ltv)
    return S.cpu().numpy(), U.cpu().numpy()


# --------------------------------------------------------------------------
# K5: the solve after the assembly (the "or")
# --------------------------------------------------------------------------
OSQP_RHO, OSQP_SIGMA, OSQP_ALPHA = 0.1, 1e-6, 1.6     # OSQP's default steps


def admm(P, q, G, h, x=None, y=None, z=None, iters=50, rho=OSQP_RHO, sigma=OSQP_SIGMA, alpha=OSQP_ALPHA,
         residuals=True, stream=None, kinv=None, kinv_valid=False):
    """``iters`` iterations of OSQP's ADMM on a batch of dense QPs ``min 1/2 x'Px + q'x s.t. Gx <= h``
    (``mpcasm_admm``) -- the solver call of the walking loop, ``osqp_solve_qp(P=Q, q=q, G=A, h=h)``
    (biped_mpc_loop.py:60), on the device tensors :meth:`Assembler.assemble` returns: ``P (B, no, no)``,
    ``q (B, no)``, ``G (B, nc, no)``, ``h (B, nc)``.  ``x, y, z``: the iterates of a warm start (all three,
    device tensors, updated IN PLACE) or None for a cold start.  Returns ``x, y, z, res`` with
    ``res (B, 2)`` = OSQP's primal and dual residuals (None when ``residuals`` is off).
    ``kinv``: a ``(B, no, no)`` device tensor for the inverse of ``P + sigma I + rho G'G`` -- written by this call,
    or, with ``kinv_valid``, read instead of factoring (``P``, ``G``, ``rho``, ``sigma`` unchanged since the call that
    wrote it: a new ``given`` on the same model changes ``q`` and ``h`` only)."""
    torch = require_device()
    for t in (P, q, G, h):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("P, q, G, h: contiguous float64 device tensors")
    if P.dim() != 3 or P.shape[1] != P.shape[2] or q.shape != P.shape[:2] or G.dim() != 3 or \
            G.shape[0] != P.shape[0] or G.shape[2] != P.shape[1] or h.shape != G.shape[:2]:
        raise ValueError("shapes: P (B, no, no), q (B, no), G (B, nc, no), h (B, nc)")
    batch, no, nc = P.shape[0], P.shape[1], G.shape[1]
    warm = x is not None or y is not None or z is not None
    if warm:
        if x is None or y is None or z is None:
            raise ValueError("a warm start takes x, y and z")
        for t, shape in ((x, (batch, no)), (y, (batch, nc)), (z, (batch, nc))):
            if not (isinstance(t, torch.Tensor) and t.device == P.device and t.dtype == torch.float64
                    and t.is_contiguous() and tuple(t.shape) == shape):
                raise ValueError("x (B, no), y (B, nc), z (B, nc): contiguous float64 tensors on P's device")
    else:
        x = torch.empty((batch, no), dtype=torch.float64, device=P.device)
        y = torch.empty((batch, nc), dtype=torch.float64, device=P.device)
        z = torch.empty((batch, nc), dtype=torch.float64, device=P.device)
    res = torch.empty((batch, 2), dtype=torch.float64, device=P.device) if residuals else None
    if kinv is not None and not (isinstance(kinv, torch.Tensor) and kinv.device == P.device and kinv.dtype == torch.float64
                                 and kinv.is_contiguous() and tuple(kinv.shape) == (batch, no, no)):
        raise ValueError("kinv: a contiguous float64 (B, no, no) tensor on P's device")
    if kinv_valid and kinv is None:
        raise ValueError("kinv_valid without kinv")
    with torch.cuda.device(P.device):
        rc = capi.load().mpcasm_admm(no, nc, P.data_ptr(), q.data_ptr(), G.data_ptr(), h.data_ptr(),
                                     x.data_ptr(), y.data_ptr(), z.data_ptr(),
                                     res.data_ptr() if residuals else None, float(rho), float(sigma),
                                     float(alpha), int(iters), 1 if warm else 0, batch,
                                     kinv.data_ptr() if kinv is not None else None, 1 if kinv_valid else 0,
                                     _stream_handle(torch, stream))
    capi.check(rc, "mpcasm_admm")
    return x, y, z, res


# --------------------------------------------------------------------------
# K2 + K3 + K4
# --------------------------------------------------------------------------
HALF_CU_LDS = 80 * 1024      # two workgroups of the persistent kernel share a CU's 160 KB


def resident_lds_bytes(plan):
    """LDS bytes a workgroup of the persistent kernel needs for ``plan`` -- with ``P`` handed over
    directly, with ``P`` collected in LDS (``mpcasm_resident_lds_bytes``; 0: not on that kernel)."""
    out = (ctypes.c_int64 * 2)()
    itab, dtab = np.ascontiguousarray(plan.itab), np.ascontiguousarray(plan.dtab)
    capi.check(capi.load().mpcasm_resident_lds_bytes(
        itab.ctypes.data, itab.size, dtab.ctypes.data if dtab.size else None, dtab.size, out),
        "mpcasm_resident_lds_bytes")
    return int(out[0]), int(out[1])


STREAMING_LAUNCH_BYTES = 560e6    # results per launch from which P is collected in LDS (resident.hip)


def plan_for_device(form, workspace="auto", batch=None, **kw):
    """``compile_plan`` plus the one decision that needs the kernel's own LDS layout: with
    ``workspace="auto"`` a plan whose dense workspace leaves room for ONE workgroup per CU is compiled
    with the compact workspace when that fits two -- judged with ``P`` handed over directly, and, for
    an assembler whose capacity makes its launches streaming ones, first with ``P`` in LDS, which is
    how such launches run (same box, one process, `tools/ab_workspace.py`: the biped at N = 24 with
    S, U read from memory 0.076 -> 0.064 ms at 4 096 instances, 0.276 -> 0.264 at 16 384; with its
    matrices built on chip 0.230 -> 0.219 at 16 384, where the dense plan would be 7 % faster at 4 096)."""
    plan = compile_plan(form, workspace=workspace, **kw)
    if workspace != "auto" or plan.workspace.compact:
        return plan
    dense = resident_lds_bytes(plan)
    if max(dense) <= HALF_CU_LDS:
        return plan
    per_instance = 8 * (plan.no * plan.no + plan.no + plan.nc * plan.no + plan.nc)
    streaming = batch is not None and per_instance * batch >= STREAMING_LAUNCH_BYTES
    small, lds = None, None
    for in_lds in ((1, 0) if streaming else (0,)):
        if dense[in_lds] > HALF_CU_LDS:
            if small is None:
                small = compile_plan(form, workspace="compact", **kw)
                lds = resident_lds_bytes(small)
            if small.workspace.compact and 0 < lds[in_lds] <= HALF_CU_LDS:
                return small
    return plan


def _checked_out(torch, out, shape, device, what):
    """A caller's result buffer: float64, contiguous, on the assembler's device, at least ``shape``."""
    if (not torch.is_tensor(out) or out.dtype != torch.float64 or not out.is_contiguous()
            or out.device != device or out.dim() != len(shape) or out.shape[1:] != tuple(shape[1:])
            or out.shape[0] < shape[0]):
        raise ValueError("%s: out must be a contiguous float64 tensor on %s of shape (>= %d, %s)"
                         % (what, device, shape[0], ", ".join(str(x) for x in shape[1:])))
    return out


class Assembler:
    """Batched assembly of one Formulation structure on one device.

    Per-instance numbers:
      * ``given``  ``(B, ng)``  -- argument of :meth:`assemble`;
      * parameters (weight / aim / cross_aim of every Cost, arrow / center /
        extreme of every Constraint) -- start as the Formulation's current
        values broadcast over the batch, override with :meth:`set_param`;
      * horizon matrices -- shared by default (the Formulation's own arrays),
        bind a ``(B, N, p, n)`` tensor with :meth:`bind_source` for per-instance
        dynamics (e.g. the output of :func:`fill_su`);
      * or, for the dynamics named in ``lti``, no horizon matrices at all: the kernel
        builds them on chip from the system's ``(A, B)`` (K1 fused into the assembly) --
        shared by default (recovered from the Formulation's ``S, U``), per instance with
        :meth:`bind_lti`.
    """

    def __init__(self, form, batch=1, device=None, costs=None, limits=None, lti=(), csc=None,
                 workspace="auto", ltv=()):
        torch = require_device()
        self._torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self.batch = int(batch)
        # csc = "upper" / "full": :meth:`assemble` returns the CSC ``data`` arrays of P (its upper
        # triangle / all of it) and G instead of the dense matrices -- ``(B, nnz)`` each, on the
        # pattern of :meth:`csc_pattern` -- written by the assembly kernel itself
        # (biped_mpc_loop.py:57-58 without a second pass).  ValueError / RuntimeError when the
        # problem does not run on the persistent kernel: assemble dense and use export_csc.
        # workspace = "auto": the persistent kernel's workspace is kept compact (plan.py Workspace)
        # when it is big (the plan compiler's rule: C3) or when that is what lets a second
        # workgroup share the CU's LDS (plan_for_device: the biped at N = 24 with S, U read from
        # memory); "dense" / "compact" force one.
        # ltv = [name]: the dynamics ``name`` has its own (A_k, B_k) at every step, per instance
        # (:meth:`bind_ltv`; BASELINE config C5): the sweep kernel assembles without horizon matrices
        self.plan = plan_for_device(form, costs=costs, limits=limits, lti=tuple(lti), csc=csc,
                                    workspace=workspace, batch=self.batch, ltv=tuple(ltv))
        self.csc = self.plan.csc
        p = self.plan
        self.ng, self.no, self.nc = p.ng, p.no, p.nc

        lib = capi.load()
        self._handle = ctypes.c_void_p()
        itab = np.ascontiguousarray(p.itab)
        dtab = np.ascontiguousarray(p.dtab)
        with torch.cuda.device(self.device):
            rc = lib.mpcasm_plan_create(
                itab.ctypes.data, itab.size, dtab.ctypes.data if dtab.size else None, dtab.size,
                ctypes.byref(self._handle))
        capi.check(rc, "mpcasm_plan_create")

        # sources: shared copies of the formulation's horizon matrices
        self._src = [_as_device(torch, s.array, self.device) for s in p.sources]
        self._src_stride = [0] * len(p.sources)
        self._src_index = {s.key: i for i, s in enumerate(p.sources)}
        self._lti = {g["name"]: g for g in p.lti}
        for g in p.lti:     # S[0][j][i] = A[i][j], U_j[0][0][i] = B[i][j]  (tools.py:14-33)
            ids = g["ids"]
            A = p.sources[ids[-1]].array[0].T
            Bm = np.stack([p.sources[ids[j]].array[0, 0, :] for j in range(g["m"])], axis=1)
            self.bind_lti(g["name"], A, Bm)

        self._ltv = {g["name"]: g for g in p.ltv}
        self._bind_ltv_nominal([s.array for s in p.sources])

        base = torch.as_tensor(p.params, dtype=torch.float64, device=self.device)
        self.params = base.unsqueeze(0).repeat(self.batch, 1).contiguous()

        self._work = None
        self._workspace()
        self._out = None
        # what a launch at this capacity would compile, now (not in the middle of a control loop)
        with torch.cuda.device(self.device):
            capi.check(lib.mpcasm_plan_prepare(self._handle, self.batch), "mpcasm_plan_prepare")
        self._csc = {}

    def __del__(self):
        handle = getattr(self, "_handle", None)
        if handle:
            try:
                capi.load().mpcasm_plan_destroy(handle)
            except Exception:
                pass
            self._handle = None

    def _workspace(self):
        """The scratch buffer of a launch, as large as the kernels the options in force pick
        ask for (``mpcasm_workspace_bytes``: a wide problem needs a few KB per instance on the
        tiled kernel, its whole preview-matrix workspace on the staged pipeline)."""
        nbytes = ctypes.c_size_t()
        capi.check(capi.load().mpcasm_workspace_bytes(self._handle, self.batch, ctypes.byref(nbytes)),
                   "mpcasm_workspace_bytes")
        need = max(nbytes.value // 8, 1)
        if self._work is None or self._work.numel() < need:
            self._work = None
            self._work = self._torch.empty(need, dtype=self._torch.float64, device=self.device)
        return self._work

    def set_option(self, option, value):
        """This assembler's own kernel path (``capi.OPT_PATH``), per-plan compilation
        (``capi.OPT_JIT``) or workgroups per CU (``capi.OPT_RESIDENT_PER_CU``); ``-1``: the
        process-wide value again.  Plan state, unlike ``mpcasm_set_option``."""
        capi.check(capi.load().mpcasm_plan_set_option(self._handle, int(option), int(value)),
                   "mpcasm_plan_set_option")

    # ---- per-instance numbers -------------------------------------------------
    def refresh_params(self):
        """Re-read weights / aims / arrows / centres / extremes from the Cost and
        Constraint objects the plan was compiled from and broadcast them over the
        batch.  Returns False when a field changed shape (recompile needed)."""
        values = self.plan.current_params()
        if values is None:
            return False
        if self.batch == 1:   # (the drop-in tick: one upload straight into place, no broadcast kernel)
            self.params.copy_(self._torch.from_numpy(
                np.ascontiguousarray(values, dtype=np.float64)).reshape(self.params.shape))
            return True
        base = self._torch.as_tensor(values, dtype=self._torch.float64, device=self.device)
        self.params[:] = base.unsqueeze(0)
        return True

    def source_keys(self):
        return list(self._src_index.keys())

    def rebind_sources(self, form, frozen=None):
        """Re-read every horizon matrix / coefficient block from ``form`` (``frozen``: the
        snapshot of the horizon matrices taken by ``make_preview_matrices``) into this
        assembler's shared source slots -- what a tick changes when the structure stays.
        Returns False when a block no longer is what the plan compiled it as (recompile)."""
        fresh = []
        for s in self.plan.sources:
            block = s.getter(form, frozen) if s.getter is not None else s.array
            if block is None:
                return False
            fresh.append(np.ascontiguousarray(block, dtype=np.float64))
        # a U_j the plan's tile masks / CSC patterns took as causal (zeros above the diagonal,
        # tools.py:27-31) no longer is: the tables do not hold for it
        if any(not is_causal(fresh[i]) for i in self.plan.causal_assumed):
            return False
        generated = {i for g in self.plan.lti + self.plan.ltv for i in g["ids"]}
        # all blocks side by side in one device arena, filled by ONE copy from a pinned staging
        # buffer (a tick of the walking loop is copy-bound: ~20 us per separate upload)
        torch = self._torch
        shared = [i for i in range(len(fresh)) if i not in generated]
        starts = np.cumsum([0] + [fresh[i].size + (fresh[i].size & 1) for i in shared])   # (16-byte aligned)
        arena = getattr(self, "_src_arena", None)
        if arena is None or arena[0].numel() != int(starts[-1]):
            arena = self._src_arena = (
                torch.empty(int(starts[-1]), dtype=torch.float64, device=self.device),
                torch.empty(int(starts[-1]), dtype=torch.float64).pin_memory())
        stage = arena[1].numpy()
        for a, i in zip(starts, shared):
            stage[a:a + fresh[i].size] = fresh[i].ravel()
        arena[0].copy_(arena[1])
        for a, i in zip(starts, shared):
            self._src[i] = arena[0][a:a + fresh[i].size].view(fresh[i].shape)
            self._src_stride[i] = 0
        for g in self.plan.lti:     # S[0][j][i] = A[i][j], U_j[0][0][i] = B[i][j]  (tools.py:14-33)
            ids = g["ids"]
            A = fresh[ids[-1]][0].T
            Bm = np.stack([fresh[ids[j]][0, 0, :] for j in range(g["m"])], axis=1)
            self.bind_lti(g["name"], A, Bm)
        self._bind_ltv_nominal(fresh)
        return True

    def bind_source(self, key, tensor, check=True):
        """Use ``tensor`` for the horizon matrix ``key = (dynamics name, k)``:
        shape ``(N, p, n)`` (shared) or ``(B, N, p, n)`` (one per instance).

        Where the plan's tables rely on the matrix being causal -- ``U_j[k][l] = 0`` for ``l > k``,
        as ``tools.extend_matrices`` / :func:`fill_su` produce it: the tile masks of the tiled kernel
        (wide problems) and the CSC patterns -- a tensor that is not raises ``ValueError`` (one
        reduction on the device per call; ``check=False`` for a caller that guarantees it)."""
        torch = self._torch
        i = self._src_index[key]
        if key[0] in self._lti or key[0] in self._ltv:
            raise ValueError("the horizon matrices of %r are generated on chip: bind_lti / bind_ltv" % key[0])
        shape = tuple(self.plan.sources[i].array.shape)
        t = _as_device(torch, tensor, self.device)
        if check and i in self.plan.causal_assumed and tuple(t.shape)[-3:] == shape:
            N = shape[0]
            above = torch.triu(torch.ones(N, N, dtype=torch.bool, device=self.device), diagonal=1)
            if bool((t[..., above, :] != 0).any().item()):
                raise ValueError(
                    "source %r: the plan was compiled for a causal horizon matrix (zeros above the "
                    "diagonal, as tools.extend_matrices produces it); compile a plan from a "
                    "formulation that holds such a matrix to bind this one" % (key,))
        if tuple(t.shape) == shape:
            self._src[i], self._src_stride[i] = t, 0
        elif tuple(t.shape) == (self.batch,) + shape:
            self._src[i], self._src_stride[i] = t, int(np.prod(shape))
        else:
            raise ValueError("source %r expects %s or %s, got %s"
                             % (key, shape, (self.batch,) + shape, tuple(t.shape)))

    def _bind_ltv_nominal(self, arrays):
        """Until :meth:`bind_ltv` says otherwise: at every step the pair the horizon matrices in
        ``arrays`` were extended from, S[0][j][i] = A[i][j], U_j[0][0][i] = B[i][j] (tools.py:14-33)."""
        keys = [s.key for s in self.plan.sources]
        for g in self.plan.ltv:
            m = g["m"]
            A = arrays[keys.index((g["name"], m))][0].T
            Bm = np.stack([arrays[keys.index((g["name"], j))][0, 0, :] for j in range(m)], axis=1)
            self.bind_ltv(g["name"], np.broadcast_to(A, (g["N"],) + A.shape).copy(),
                          np.broadcast_to(Bm, (g["N"],) + Bm.shape).copy())

    def bind_ltv(self, name, A, B):
        """Per-step system matrices of a dynamics compiled as ``ltv``: ``A`` ``(N, n, n)`` /
        ``(B, N, n, n)`` and ``B`` ``(N, n, m)`` / ``(B, N, n, m)``, ``x_{k+1} = A_k x_k + B_k u_k``.
        They travel in the slots of the dynamics' first two horizon matrices (include/mpcasm.h)."""
        torch = self._torch
        g = self._ltv[name]
        n, m, N = g["n"], g["m"], g["N"]
        for slot, (t, shape) in enumerate(((A, (N, n, n)), (B, (N, n, m)))):
            t = _as_device(torch, t, self.device)
            i = g["ids"][slot]
            if tuple(t.shape) == shape:
                self._src[i], self._src_stride[i] = t, 0
            elif tuple(t.shape) == (self.batch,) + shape:
                self._src[i], self._src_stride[i] = t, int(np.prod(shape))
            else:
                raise ValueError("%s of %r expects %s or %s, got %s"
                                 % ("AB"[slot], name, shape, (self.batch,) + shape, tuple(t.shape)))

    def bind_lti(self, name, A, B):
        """System matrices of a dynamics compiled as ``lti``: ``A`` ``(n, n)`` / ``(B, n, n)``
        and ``B`` ``(n, m)`` / ``(B, n, m)``, x+ = A x + B u.  They travel in the slots of
        the group's first two horizon matrices (include/mpcasm.h)."""
        torch = self._torch
        g = self._lti[name]
        n, m = g["n"], g["m"]
        for slot, (t, shape) in enumerate(((A, (n, n)), (B, (n, m)))):
            t = _as_device(torch, t, self.device)
            i = g["ids"][slot]
            if tuple(t.shape) == shape:
                self._src[i], self._src_stride[i] = t.contiguous(), 0
            elif tuple(t.shape) == (self.batch,) + shape:
                self._src[i], self._src_stride[i] = t.contiguous(), shape[0] * shape[1]
            else:
                raise ValueError("%s of %r expects %s or %s, got %s" % (
                    "AB"[slot], name, shape, (self.batch,) + shape, tuple(t.shape)))

    def param_slice(self, kind, name, field):
        """Columns of :attr:`params` holding one field, e.g.
        ``("cost", "track vel_x", "aim")`` or ``("limit", 3, "center")``."""
        start, rows, cols = self.plan.param_slots[(kind, name, field)]
        return slice(start, start + rows * cols), (rows, cols)

    def set_param(self, kind, name, field, values):
        """``values``: shape ``(rows, cols)`` (all instances) or ``(B, rows, cols)``."""
        torch = self._torch
        sl, (rows, cols) = self.param_slice(kind, name, field)
        v = _as_device(torch, values, self.device).reshape(-1, rows * cols)
        if v.shape[0] not in (1, self.batch):
            raise ValueError("expected 1 or %d instances, got %d" % (self.batch, v.shape[0]))
        self.params[:, sl] = v

    # ---- launches ---------------------------------------------------------------
    def _src_args(self):
        # (rebuilt only when a source has been bound to another tensor since the last launch)
        key = tuple(t.data_ptr() for t in self._src) + tuple(self._src_stride)
        if getattr(self, "_src_key", None) != key:
            n = len(self._src)
            ptrs = (ctypes.c_void_p * max(n, 1))(*key[:n])
            strides = (ctypes.c_int64 * max(n, 1))(*self._src_stride)
            self._src_key, self._src_ctypes = key, (ptrs, strides)
        return self._src_ctypes

    def assemble(self, given=None, out=None, stream=None, want_cost=True, want_constraints=True,
                 count=None, index=None, params=None):
        """Assemble the batch; returns ``(P, q, G, h)`` device tensors (``None`` for a
        skipped half).  ``given``: ``(B, ng)`` or ``(ng,)``.  ``count`` < batch assembles
        only the first ``count`` instances (buffers keep their full capacity).
        ``index`` (int32 device tensor, ``count`` entries): instance ``b`` reads row ``index[b]`` of
        ``given`` -- which may then have any number of rows, e.g. a whole fleet's -- and row ``b`` of
        everything else (``mpcasm_assemble_indexed``; a gather in front where the plan does not run
        on the persistent kernel).  ``params``: a ``(B, n_params)`` tensor to read the parameters
        from instead of :attr:`params` (a fleet keeps one per place in its step cycle)."""
        torch = self._torch
        B, ng, no, nc = self.batch, self.ng, self.no, self.nc
        n_run = B if count is None else int(count)
        if not 0 <= n_run <= B:
            raise ValueError("count must lie in [0, %d]" % B)
        if index is not None:
            if (not torch.is_tensor(index) or index.dtype != torch.int32 or index.device != self.device
                    or not index.is_contiguous() or index.numel() < n_run or not ng):
                raise ValueError("index: a contiguous int32 tensor of %d entries on %s" % (n_run, self.device))
            g = _as_device(torch, given, self.device).reshape(-1, ng)
        elif ng:
            g = _as_device(torch, given, self.device).reshape(-1, ng)
            if g.shape[0] == 1 and B > 1:
                g = g.repeat(B, 1)
            if g.shape[0] < n_run or (count is None and g.shape[0] != B):
                raise ValueError("given must have %d rows, got %d" % (n_run, g.shape[0]))
        else:
            g = None
        if params is not None and (tuple(params.shape) != tuple(self.params.shape) or params.dtype != torch.float64
                                   or params.device != self.device or not params.is_contiguous()):
            raise ValueError("params: a contiguous float64 tensor of shape %s on %s"
                             % (tuple(self.params.shape), self.device))
        if out is None:
            if self._out is None:
                f = dict(dtype=torch.float64, device=self.device)
                pshape = (B, self.csc["pnnz"]) if self.csc else (B, no, no)
                gshape = (B, self.csc["gnnz"]) if self.csc else (B, nc, no)
                self._out = (torch.empty(pshape, **f), torch.empty((B, no), **f),
                             torch.empty(gshape, **f), torch.empty((B, nc), **f))
            out = self._out
        P, q, G, h = out
        if not want_cost:
            P = q = None
        if not want_constraints or nc == 0:
            G = h = None
        self._launch(g, (P, q, G, h), n_run, stream, index, params)
        return P, q, G, h

    def _launch(self, g, out, n_run, stream, index=None, params=None):
        torch = self._torch
        ptrs, strides = self._src_args()
        ptr = lambda t: t.data_ptr() if t is not None else None
        work = self._workspace()
        prm = self.params if params is None else params
        with torch.cuda.device(self.device):
            if index is not None:
                rc = capi.load().mpcasm_assemble_indexed(
                    self._handle, ptrs, strides, prm.data_ptr(), ptr(g), index.data_ptr(),
                    *(ptr(t) for t in out), work.data_ptr(), n_run, _stream_handle(torch, stream))
                if rc == capi.ERR_LIMIT:     # (not on the persistent kernel: gather, then as ever)
                    g = g.index_select(0, index[:n_run].long())
                    index = None
            if index is None:
                rc = capi.load().mpcasm_assemble(
                    self._handle, ptrs, strides, prm.data_ptr(), ptr(g), *(ptr(t) for t in out),
                    work.data_ptr(), n_run, _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_assemble")

    def last_kernel(self):
        """Name of the kernel(s) the latest :meth:`assemble` launched (``mpcasm_plan_last_kernel``)."""
        return capi.KERNEL_NAMES.get(capi.load().mpcasm_plan_last_kernel(self._handle), "?")

    # ---- sparse hand-off (f3) ------------------------------------------------------
    def csc_pattern(self, which, upper=False):
        """``(indptr, indices)`` (numpy int32) of the batch-wide CSC pattern of ``"P"`` or
        ``"G"``: every entry that can be non-zero for this structure, whatever the numbers
        (``upper``: only the upper triangle, as OSQP wants P).  Dense when the plan is too
        large for the structural analysis."""
        from .plan import csc_pattern

        if self.csc:      # the pattern the assembly itself writes
            return self.csc[which]
        key = (which, bool(upper))
        if key not in self._csc:
            mask = {"P": self.plan.P_pattern, "G": self.plan.G_pattern}[which]
            if mask is None:
                mask = np.ones((self.no, self.no) if which == "P" else (self.nc, self.no), dtype=bool)
            indptr, indices, flat = csc_pattern(mask, upper)
            self._csc[key] = (indptr, indices,
                              self._torch.as_tensor(flat, dtype=self._torch.int32, device=self.device))
        return self._csc[key][0], self._csc[key][1]

    def export_csc(self, which, dense=None, upper=False, count=None, stream=None):
        """``(B, nnz)`` device tensor: the ``data`` arrays of ``scipy.sparse.csc_matrix(M)`` for
        every instance's ``M = P`` or ``G`` on the pattern of :meth:`csc_pattern`
        (biped_mpc_loop.py:57-58, batched).  ``dense`` defaults to the last :meth:`assemble`."""
        torch = self._torch
        if self.csc:
            raise ValueError("this assembler writes the CSC form itself: assemble() returns it")
        self.csc_pattern(which, upper)
        index = self._csc[(which, bool(upper))][2]
        if dense is None:
            dense = self._out[0 if which == "P" else 2]
        n = self.batch if count is None else int(count)
        out = torch.empty((n, index.numel()), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_gather(
                dense.data_ptr(), dense[0].numel(), index.data_ptr(), index.numel(),
                out.data_ptr(), n, _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_gather")
        return out

    def preview_matrices(self, stream=None):
        """``(B, preview_rows, ng+no)`` device tensor; rows of definition ``v`` are
        ``plan.pm_rows[v]``, columns ``[:ng]`` = Mg, ``[ng:]`` = Mo."""
        torch = self._torch
        W = self.ng + self.no
        PM = torch.empty((self.batch, self.plan.pmrows, W), dtype=torch.float64,
                         device=self.device)
        ptrs, strides = self._src_args()
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_preview_matrices(
                self._handle, ptrs, strides, PM.data_ptr(), self.batch,
                _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_preview_matrices")
        return PM

    def preview_rows(self, given, optim, out=None, stream=None, count=None):
        """``Mg @ given + Mo @ optim`` for every row of every definition (body.py:209-219)
        without a preview matrix in memory (``mpcasm_preview_direct``): ``(B, preview_rows)``;
        rows of definition ``v`` are ``plan.pm_rows[v]``.  ``given``: ``(B, ng)``, ``optim``:
        ``(B, no)`` (the solver's answer)."""
        torch = self._torch
        n = self.batch if count is None else int(count)
        if not 0 <= n <= self.batch:         # (the scratch and the result are sized for the batch)
            raise ValueError("count must lie in 0 .. %d, got %d" % (self.batch, n))
        g = _as_device(torch, given, self.device).reshape(-1, self.ng) if self.ng else None
        x = _as_device(torch, optim, self.device).reshape(-1, self.no) if self.no else None
        for t, name in ((g, "given"), (x, "optim")):
            if t is not None and t.shape[0] < n:
                raise ValueError("%s must have %d rows, got %d" % (name, n, t.shape[0]))
        if out is None:
            out = torch.empty((self.batch, self.plan.pmrows), dtype=torch.float64, device=self.device)
        else:
            _checked_out(torch, out, (n, self.plan.pmrows), self.device, "preview_rows")
        ptrs, strides = self._src_args()
        work = self._workspace()
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_preview_direct(
                self._handle, ptrs, strides, g.data_ptr() if g is not None else None,
                x.data_ptr() if x is not None else None, out.data_ptr(), work.data_ptr(), n,
                _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_preview_direct")
        return out

    def goal_terms(self, form):
        """Device table of ``mpcasm_goal_distance`` for ``form.goals`` (the plan's costs): one
        record ``(goal, first preview row, rows, aim's parameter slot)`` per goal and axis;
        returns ``(table, goal names)``."""
        if getattr(self, "_goal_terms", None) is None:
            names, recs = list(form.goals.keys()), []
            for gi, name in enumerate(names):
                goal = form.goals[name]
                aim0 = self.plan.param_slots[("cost", name, "aim")][0]
                for i, axis in enumerate(goal.axes):
                    r0, rows = self.plan.pm_rows[goal.variable + axis]
                    recs.append([gi, r0, rows, aim0 + i])
            table = self._torch.as_tensor(np.asarray(recs, dtype=np.int32).reshape(-1, 4),
                                          device=self.device)
            self._goal_terms = (table, names)
        return self._goal_terms

    def goal_distance(self, form, rows, out=None, stream=None, count=None):
        """Squared distance of every goal's variable to its aim (body.py:221-228), per instance:
        ``(B, n_goals)`` from the rows of :meth:`preview_rows`; the row sum over the goals is
        ``full_goal_distance`` (:230-234).  Aims are read from :attr:`params`."""
        torch = self._torch
        table, names = self.goal_terms(form)
        n = self.batch if count is None else int(count)
        if not 0 <= n <= self.batch or rows.shape[0] < n:
            raise ValueError("count must lie in 0 .. %d and within the %d rows given, got %d"
                             % (self.batch, rows.shape[0], n))
        if out is None:
            out = torch.empty((self.batch, len(names)), dtype=torch.float64, device=self.device)
        else:
            _checked_out(torch, out, (n, len(names)), self.device, "goal_distance")
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_goal_distance(
                rows.data_ptr(), rows.shape[1], self.params.data_ptr(), self.params.shape[1],
                table.data_ptr(), table.shape[0], len(names), out.data_ptr(), n,
                _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_goal_distance")
        return out

    def full_goal_distances(self, form, given, optim, out=None, stream=None, count=None):
        """``goal_distance`` of every goal (body.py:221-234) straight from the sources, the rows
        of the definitions never leaving the chip (``mpcasm_preview_goal_distance``): ``(B, n_goals)``;
        the row sum is ``full_goal_distance``.  Falls back to :meth:`preview_rows` +
        :meth:`goal_distance` where the plan does not run on the kernel that fuses the two."""
        torch = self._torch
        table, names = self.goal_terms(form)
        n = self.batch if count is None else int(count)
        if not 0 <= n <= self.batch:
            raise ValueError("count must lie in 0 .. %d, got %d" % (self.batch, n))
        g = _as_device(torch, given, self.device).reshape(-1, self.ng) if self.ng else None
        x = _as_device(torch, optim, self.device).reshape(-1, self.no) if self.no else None
        for t, name in ((g, "given"), (x, "optim")):
            if t is not None and t.shape[0] < n:
                raise ValueError("%s must have %d rows, got %d" % (name, n, t.shape[0]))
        if out is None:
            out = torch.empty((self.batch, len(names)), dtype=torch.float64, device=self.device)
        else:
            _checked_out(torch, out, (n, len(names)), self.device, "full_goal_distances")
        ptrs, strides = self._src_args()
        work = self._workspace()
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_preview_goal_distance(
                self._handle, ptrs, strides, g.data_ptr() if g is not None else None,
                x.data_ptr() if x is not None else None, self.params.data_ptr(), table.data_ptr(),
                table.shape[0], len(names), out.data_ptr(), work.data_ptr(), n, _stream_handle(torch, stream))
        if rc == capi.ERR_LIMIT:
            rows = self.preview_rows(given, optim, stream=stream, count=count)
            return self.goal_distance(form, rows, out=out, stream=stream, count=count)
        capi.check(rc, "mpcasm_preview_goal_distance")
        return out

    def preview(self, PM, given, optim, stream=None):
        """``Mg @ given + Mo @ optim`` for every definition row (body.py:209-219):
        ``(B, preview_rows)``."""
        torch = self._torch
        g = _as_device(torch, given, self.device).reshape(self.batch, self.ng) if self.ng else None
        x = _as_device(torch, optim, self.device).reshape(self.batch, self.no) if self.no else None
        out = torch.empty((self.batch, self.plan.pmrows), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = capi.load().mpcasm_preview(
                PM.data_ptr(), g.data_ptr() if g is not None else None,
                x.data_ptr() if x is not None else None, out.data_ptr(), self.batch,
                self.plan.pmrows, self.ng, self.no, _stream_handle(torch, stream))
        capi.check(rc, "mpcasm_preview")
        return out
