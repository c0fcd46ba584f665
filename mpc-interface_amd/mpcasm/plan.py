"""Plan compiler: Formulation structure -> flat device tables.

The reference re-interprets its dict-of-dicts problem description on every tick
(body.py:149-193 for the preview matrices, :236-329 for the QP blocks).  Here
the *structure* -- index maps, the flattened definition graph, which rows every
cost / constraint consumes -- is compiled once into two flat arrays (``itab``
int32, ``dtab`` float64, layout in csrc/plan_tables.h) and the *numbers* that
differ between instances of a batch (given vector, weights, aims, arrows,
centres, extremes, horizon matrices) stay outside, as device buffers.

This module is host logic only (numpy / scipy.sparse); it never evaluates a
preview matrix or a QP block -- that is the kernels' job.
"""
import hashlib

import numpy as np
import scipy.sparse as sp

PLAN_MAGIC = 0x4D504341
PLAN_VERSION = 30

# header words (csrc/plan_tables.h, enum HeaderWord)
_H = {name: i for i, name in enumerate([
    "MAGIC", "VERSION", "NG", "NO", "NC", "NPARAMS", "NSRC", "NBASE", "NSEG", "RTOT", "NENT",
    "NGTERM", "NLIMIT", "NLAX", "PMROWS", "PM_NENT", "LDV",
    "OFF_SEG", "OFF_COLSEG", "OFF_ROWPTR", "OFF_ENTBASE", "OFF_ENTK", "OFF_GTERM", "OFF_LIMIT",
    "OFF_LAX", "OFF_ROWLIMIT", "OFF_PM_ROWPTR", "OFF_PM_ENTBASE", "OFF_PM_ENTK",
    "DOFF_ENTCOEF", "DOFF_PM_ENTCOEF", "NITAB", "NDTAB",
    "FUSED_OK", "ARENA_TOTAL", "OFF_ARENA", "NFD", "OFF_FD_IDX", "OFF_FD_PTR", "NOPS", "OFF_OP",
    "NCOEF", "DOFF_COEFPOOL",
    "RS_OK", "RS_JC", "RS_SYM", "RS_NTRIP", "OFF_RS_SRC", "OFF_RS_GIDX", "OFF_RS_DST",
    "DOFF_RS_COEF", "OFF_RS_TRIP", "OFF_RS_WTRIP", "RS_NSPLIT", "OFF_RS_SPLIT",
    "OFF_RS_RR", "RS_UNIT", "RS_NCHUNK", "OFF_RS_INMETA", "RS_IMG", "RS_IMG_GIVEN",
    "RS_IMG_PARAMS", "DOFF_RS_CONST", "DOFF_DIAGCOEF", "NDIAGCOEF", "RS_NLTI", "OFF_RS_LTI",
    "RS_IMG_DMA", "RS_AB", "OFF_RS_ABMETA", "RR_PACKED", "OFF_RS_DPAR", "DOFF_RS_DCOEF",
    "RS_NGDESC", "OFF_RS_GDESC", "PM_NFD", "OFF_PM_MAP", "OFF_PM_FDPTR", "OFF_PM_OP", "PM_NOPS",
    "DOFF_PM_POOL", "PM_NPOOL", "RS_NZBLK", "OFF_RS_ZBLK", "RS_GSINGLE",
    "CSC_PNNZ", "OFF_CSC_P", "CSC_GNNZ", "OFF_CSC_G", "CSC_GSINGLE",
    "T_CI_OK", "T_NOP", "OFF_T_CIG", "OFF_T_CIO", "T_DOFF_DELTA", "T_NDELTA",
    "T_OK", "T_NSTAGE", "OFF_T_STAGE", "T_NLTI", "OFF_T_LTI", "OFF_T_LTI_IDS", "T_WORK",
    "OFF_T_GROW", "OFF_T_SROW", "T_DOFF_SCOEF", "OFF_T_PIG", "T_NGREST", "OFF_T_GREST", "OFF_T_BROW0", "OFF_T_BCOLPTR", "OFF_T_BCOLS", "T_TOEPLITZ",
    "RS_NGFIX", "OFF_RS_GFIX", "RS_COMPACT", "RS_LDV", "RS_VD", "RS_VROW0", "OFF_RS_RRWIN",
    "T_NP1", "OFF_T_P1PTR", "OFF_T_P1ENT", "OFF_T_P2Y",
    "T_SCAN", "T_SCAN_NBLK", "OFF_T_SCAN_BLK", "OFF_T_SCAN_GT", "T_DOFF_SCAN_GC", "OFF_T_SCAN_GROW",
    "T_DOFF_SCAN_GCOEF", "T_SCAN_NGREST", "OFF_T_SCAN_GREST", "OFF_T_SCAN_COLBLK", "T_SCAN_NOTHER",
    "SW_OK", "SW_N", "SW_M", "SW_HORIZON", "SW_SRC_A", "SW_SRC_B", "SW_NAXES", "OFF_SW_AXIS", "SW_NTERM",
    "OFF_SW_TERM", "SW_NLIM", "OFF_SW_LIM", "OFF_SW_COL", "SW_DOFF_CVEC", "SW_NCVEC",
    "OFF_SW_CPTR", "OFF_SW_CENT", "SW_NCENT", "OFF_SW_GPTR", "OFF_SW_GENT", "SW_NGENT",
    "OFF_RS_PROG", "T_SCAN_FUSED",
])}
H_WORDS = 160
assert len(_H) <= H_WORDS
RS_NW, RS_NT = 4, 512                     # matrix wavefronts (they fetch the inputs), threads per instance
RS_WAVES = RS_NT // 64
RS_BLOCKS_MAX = 255                       # 4-column blocks of the unknowns, a byte each
RS_LTI_WORDS, RS_LTI_MAX = 8, 4           # record of a source group generated on chip; groups per plan
RS_JC_MAX = 12                            # compose ops a thread can keep in registers
RS_TRIP_WORDS = 8
# cost model of the wavefront assignment (rounded cycles of one wavefront), see _resident_program
TRIP_COST, TRIP_STEP_COST, TERM_COST, TRIP_Q_COST, PACK_COST, G_PIECE_COST = 200, 30, 60, 60, 250, 500
G_DESC_PIECE_COST = 330     # ... a piece of G by the descriptor table (small problems; stamps: 1 950 cycles for six)
TABLES_COST, STREAM_WAVE_COST, FETCH_CHUNK_COST = 3300, 500, 350   # horizon tables on chip; h and the rest of a stream wave
import os as _os
if _os.environ.get("MPCASM_TRIP_COSTS"):          # tuning aid: "trip,step,term,q,pack,piece[,tables,stream,chunk]"
    _c = [int(x) for x in _os.environ["MPCASM_TRIP_COSTS"].split(",")]
    TRIP_COST, TRIP_STEP_COST, TERM_COST, TRIP_Q_COST, PACK_COST, G_PIECE_COST = _c[:6]
    G_DESC_PIECE_COST = G_PIECE_COST
    if len(_c) > 6:
        TABLES_COST, STREAM_WAVE_COST, FETCH_CHUNK_COST = _c[6:9]
# trip record (csrc/plan_tables.h RT_*): words A, B, D, W, AIM, WORD, BI, BJ;
# WORD = rows (16 or 4) | short << 5 | half << 6 | nop << 7 | first << 8 | last << 9 | live << 10
#        | qmask << 14 | last trip of its term in the pack << 18
RT_A, RT_B, RT_D, RT_W, RT_AIM, RT_WORD, RT_BI, RT_BJ = range(8)
RT_SHORT, RT_HALF, RT_NOP, RT_FIRST, RT_LAST, RT_LIVE, RT_QMASK, RT_TERM_END = 5, 6, 7, 8, 9, 10, 14, 18
RS_DIAG_MAX = 2                           # diagonal gterms per column (persistent kernel)
RS_AXMAX = 4                              # axes per constraint row record
RS_DST_ACC = 1 << 30                      # compose destination shared by two threads
RS_GDESC_PIECES, RS_GDESC_THREADS = 6, 256   # descriptor table of G: pieces per stream-wave thread
RS_GFIX_NONE = 7
RS_COMPACT_FROM_BYTES = 65536           # a dense workspace beyond this is compacted (tools/ab_workspace.py)
RS_RR_WORDS = 16                          # row record: voff[4], arrow param[4], center param[4], naxes, extreme param, pad
SEG_WORDS, GT_WORDS, LM_WORDS, LX_WORDS = 8, 10, 12, 2
FUSED_MAX_OPS = 1 << 18           # beyond this the staged pipeline is used
FUSED_MAX_ARENA = 1 << 14         # doubles
SEG_GATHER, SEG_IDENTITY = 0, 1
GT_FLAG_P, GT_FLAG_HALF, GT_FLAG_DIAG = 1, 2, 4
MAX_SOURCES = 32
# tiled kernel (csrc/tiled.hip, plan_tables.h T_* / TS_* / TL_*)
T_BLOCK, T_SID_CONST, T_STAGE_WORDS, T_LTI_WORDS = 128, 32, 16, 8
TS_FLAG_P, TS_FLAG_HALF, TS_FLAG_SIMPLE_A, TS_FLAG_SIMPLE_B, TS_FLAG_SAME, TS_FLAG_G, TS_FLAG_TOEPLITZ = 1, 2, 4, 8, 16, 32, 64
T_PIG_MAX = 2
# scan form of the tiled kernel (csrc/tiled.hip toeplitz_scan_kernel, plan_tables.h T_SCAN*)
T_SCAN_KMAX, T_SCAN_BLKMAX, T_SCAN_NMAX, T_SCAN_GT_WORDS = 16, 8, 64, 4
# sweep kernel (csrc/sweep.hip, plan_tables.h SW_*): per-step dynamics, the Hessian by two recursions
SW_NMAX, SW_MMAX, SW_AXMAX, SW_AXIS_WORDS, SW_TERM_WORDS, SW_LIM_WORDS, SW_LAX_WORDS = 4, 4, 4, 8, 8, 8, 8


class Source:
    """One horizon matrix read by the definitions: ``ExtendedSystem.matrices[k]``
    (key ``(dynamics name, k)``) or an anonymous constant coefficient block."""

    def __init__(self, key, array, getter=None):
        self.key = key
        self.array = np.ascontiguousarray(array, dtype=np.float64)
        # getter(form, frozen) -> the block's current numbers (same shape), or None when it is
        # no longer what the plan compiled it as; ``frozen``: {(dynamics, k): array} snapshot of
        # the horizon matrices (Formulation.make_preview_matrices), may be None
        self.getter = getter


class Plan:
    """Compiled structure of one Formulation (see module docstring)."""

    def __init__(self):
        self.itab = None
        self.dtab = None
        self.ng = self.no = self.nc = 0
        self.sources = []          # list[Source]
        self.params = None         # base parameter vector (current numbers of the form)
        self.param_slots = {}      # (kind, name, field) -> (start, rows, cols)
        self.pm_rows = {}          # definition -> (row0, rows) in the preview program
        self.pmrows = 0
        self.rtot = 0
        self.ldv = 0
        self.limit_rows = []       # (out_row0, nrows) of every limit, stacking order
        self.optim_ID = {}
        self.given_ID = {}
        self.param_getters = []    # (start, size, callable) -> current numbers of the objects
        self.fingerprint = None    # structure of the costs / limits this plan was built from
        self.causal_assumed = []   # ids of the sources whose zeros above the diagonal the tables rely on

    def current_params(self):
        """Parameter vector re-read from the Cost / Constraint objects, or ``None``
        when a field no longer has the shape it was compiled with."""
        out = np.empty_like(self.params)
        for start, size, getter in self.param_getters:
            values = np.asarray(getter(), dtype=np.float64).ravel()
            if values.size != size:
                return None
            out[start:start + size] = values
        return out

    def signature(self):
        """Bytes identifying the structure (cache key)."""
        return self.itab.tobytes() + self.dtab.tobytes()


# --------------------------------------------------------------------------
def _columns(form):
    """Column offset of every domain variable in [given | optim] (body.py:164-169)."""
    ng = form.given_len
    cols = {}
    for var in form.given_variables:
        cols[var] = (form.given_ID[var].start, len(form.given_ID[var]))
    for var in form.optim_variables:
        cols[var] = (ng + form.optim_ID[var].start, len(form.optim_ID[var]))
    return cols


class _Builder:
    def __init__(self, form):
        self.form = form
        self.ng, self.no = form.given_len, form.optim_len
        self.W = self.ng + self.no
        self.cols = _columns(form)
        self.sources = []
        self._source_ids = {}
        self.segments = []            # SEG_WORDS ints each
        self.base_ids = {}            # base variable -> id
        self.base_rows = []           # rows of every base variable
        self.base_row0 = []           # offset in the global base-row space
        self.colseg = []              # per base: array [W] of segment ids
        self.var_matrix = {}          # definition -> sparse (rows x total_base_rows)
        self.rowset_rows = []         # list of sparse row blocks
        self.rowset_keys = {}
        self.rtot = 0
        self.params = []
        self.param_slots = {}
        self.param_getters = []

    # ---- sources ---------------------------------------------------------
    def source_id(self, key, array, getter=None):
        if key not in self._source_ids:
            if len(self.sources) >= MAX_SOURCES:
                raise ValueError("more than %d horizon matrices in one formulation" % MAX_SOURCES)
            self._source_ids[key] = len(self.sources)
            self.sources.append(Source(key, array, getter))
        return self._source_ids[key]

    # ---- base variables (body.py:158-177) ----------------------------------
    def add_base(self, var):
        form = self.form
        dyn_name = form.of[var]
        dyn = form.dynamics[dyn_name]
        rows = int(dyn.all_variables[var])
        bid = len(self.base_rows)
        self.base_ids[var] = bid
        self.base_row0.append(sum(self.base_rows))
        self.base_rows.append(rows)
        colseg = -np.ones(self.W, dtype=np.int32)

        state_ID = getattr(dyn, "state_ID", {})
        for pos, (dep, matrix) in enumerate(form.definitions[var].items()):
            if dep not in self.cols:
                raise ValueError(
                    "The variable {} in the definition of {} seems to not be given nor "
                    "optimal.".format(dep, var))
            dst0, length = self.cols[dep]
            matrix = np.asarray(matrix, dtype=np.float64)
            seg = None
            if var in state_ID and dep in dyn.domain_ID:
                dID, sID = dyn.domain_ID[dep], state_ID[var]
                src = dyn.matrices[dID]
                if (src.ndim == 3 and sID < src.shape[2]
                        and matrix.shape == src.shape[:2]
                        and np.array_equal(matrix, src[..., sID])):
                    def current(form, frozen, name=dyn_name, k=dID, shape=src.shape):
                        M = (frozen or {}).get((name, k))
                        if M is None:
                            M = form.dynamics[name].matrices[k]
                        return M if np.shape(M) == shape else None

                    sid = self.source_id((dyn_name, dID), src, current)
                    n = src.shape[2]
                    seg = [sid, sID, src.shape[1] * n, n, dst0, length, SEG_GATHER, 0]
            if seg is None:
                block = np.broadcast_to(matrix, (rows, length)) if matrix.ndim < 2 else matrix
                if block.shape != (rows, length):
                    raise ValueError(
                        "coefficient of {} in the definition of {} has shape {} instead of "
                        "{}".format(dep, var, block.shape, (rows, length)))
                if rows == length and np.array_equal(block, np.eye(rows)):
                    seg = [0, 0, 0, 0, dst0, length, SEG_IDENTITY, 0]
                else:
                    def current(form, frozen, v=var, k=pos, shape=(rows, length)):
                        M = np.asarray(list(form.definitions[v].values())[k], dtype=np.float64)
                        M = np.broadcast_to(M, shape) if M.ndim < 2 else M
                        return M if M.shape == shape else None

                    sid = self.source_id(("const", var, pos), np.array(block), current)
                    seg = [sid, 0, length, 1, dst0, length, SEG_GATHER, 0]
            colseg[dst0:dst0 + length] = len(self.segments)
            self.segments.append(seg)
        self.colseg.append(colseg)

    def lti_state_of(self, group):
        """``{base id: state index}`` of the bases that are states of the generated group: every
        segment of such a base reads one of the group's sources at element offset = its state."""
        ids = set(group["ids"])
        out = {}
        for var, bid in self.base_ids.items():
            segs = [self.segments[sg] for sg in sorted(set(int(x) for x in self.colseg[bid] if x >= 0))]
            gather = [sg for sg in segs if sg[6] == SEG_GATHER]
            if gather and len(gather) == len(segs) and all(sg[0] in ids for sg in gather):
                states = {sg[1] for sg in gather}
                if len(states) == 1:
                    out[bid] = states.pop()
        return out

    # ---- flattened definition graph (body.py:179-193) ----------------------
    def flatten_definitions(self):
        form = self.form
        for var in form.definitions.keys():
            if var in form.of:
                self.add_base(var)
        total = sum(self.base_rows)
        for var, combo in form.definitions.items():
            if var in form.of:
                bid = self.base_ids[var]
                rows, r0 = self.base_rows[bid], self.base_row0[bid]
                M = sp.csr_matrix(
                    (np.ones(rows), (np.arange(rows), r0 + np.arange(rows))), shape=(rows, total))
            else:
                M = None
                for dep, coef in combo.items():
                    D = self.var_matrix[dep]
                    c = np.array(coef, dtype=np.float64)
                    if c.ndim == 0:
                        term = D * float(c)
                    elif c.ndim == 1:
                        term = sp.csr_matrix(c[None, :]) @ D
                    else:
                        term = sp.csr_matrix(c) @ D
                    M = term if M is None else M + term
                M = sp.csr_matrix(M)
            M.sum_duplicates()
            M.sort_indices()
            self.var_matrix[var] = M
        self.total_base_rows = total

    # ---- row-sets -----------------------------------------------------------
    def rowset(self, var, schedule, L):
        """Rows of ``L @ M_var[schedule]`` (or ``M_var[schedule]``); returns
        ``(row-set id, nrows)``.  ``schedule`` falsy = all rows.  Workspace offsets are
        handed out later, only to the row-sets some kernel really has to read
        (:meth:`place_rowsets`)."""
        M = self.var_matrix[var]
        if schedule:
            pick = list(schedule)
            key_s = (schedule.start, schedule.stop, schedule.step)
        else:
            pick, key_s = None, None
        if L is not None:
            Lm = np.atleast_2d(np.asarray(L, dtype=np.float64))
            key_l = (Lm.shape, Lm.tobytes())
        else:
            Lm, key_l = None, None
        key = (var, key_s, key_l)
        if key in self.rowset_keys:
            return self.rowset_keys[key]
        block = M[pick] if pick is not None else M
        if Lm is not None:
            if Lm.shape[1] != block.shape[0]:
                raise ValueError(
                    "L with {} columns applied to {} rows of {}".format(
                        Lm.shape[1], block.shape[0], var))
            block = sp.csr_matrix(sp.csr_matrix(Lm) @ block)
        block = sp.csr_matrix(block)
        block.sum_duplicates()
        block.sort_indices()
        out = (len(self.rowset_rows), block.shape[0])
        self.rowset_rows.append(block)
        self.rowset_keys[key] = out
        return out

    def diagonal_rowset(self, rid):
        """``(first optim column, coefficients)`` when row ``r`` of the row-set is
        ``coef[r] * e_{col0 + r}`` on the unknowns and has no given part (a cost on a
        free variable itself, e.g. the jerk of the LIPM); else ``None``."""
        block = self.rowset_rows[rid]
        n = block.shape[0]
        if n == 0 or block.nnz != n or np.any(np.diff(block.indptr) != 1):
            return None
        starts = np.asarray(self.base_row0)
        cols, coefs = [], []
        for r in range(n):
            gidx = int(block.indices[r])
            u = int(np.searchsorted(starts, gidx, side="right") - 1)
            k = gidx - int(starts[u])
            segs = sorted(set(int(x) for x in self.colseg[u] if x >= 0))
            if len(segs) != 1:
                return None
            seg = self.segments[segs[0]]
            if seg[6] != SEG_IDENTITY or seg[4] < self.ng or k >= seg[5]:
                return None
            cols.append(seg[4] - self.ng + k)
            coefs.append(float(block.data[r]))
        if any(c != cols[0] + i for i, c in enumerate(cols)):
            return None
        return cols[0], np.asarray(coefs)

    def place_rowsets(self, needed):
        """Workspace row offset of every needed row-set (id order, each a multiple of four);
        the others are never materialised."""
        offsets, blocks = {}, []
        self.rtot = 0
        self.spans = []                      # (first row, rows incl. the zero rows behind them)
        for rid, block in enumerate(self.rowset_rows):
            if rid in needed:
                offsets[rid] = self.rtot
                self.spans.append((self.rtot, block.shape[0] + -block.shape[0] % 4))
                self.rtot += block.shape[0]
                blocks.append(block)
                pad = -block.shape[0] % 4
                if pad:
                    # every row-set starts a group of four rows and is followed by zero rows up
                    # to the next one: the matrix core then takes the rows of a term four at a
                    # time without masking (rows nothing composes stay exact zeros)
                    blocks.append(sp.csr_matrix((pad, block.shape[1])))
                    self.rtot += pad
        self.placed_blocks = blocks
        return offsets

    # ---- parameters -----------------------------------------------------------
    def param(self, key, getter):
        values = np.atleast_2d(np.asarray(getter(), dtype=np.float64))
        start = len(self.params)
        self.params.extend(values.ravel().tolist())
        self.param_slots[key] = (start, values.shape[0], values.shape[1])
        self.param_getters.append((start, values.size, getter))
        return start


def _digest(a):
    """Shape and a digest of the numbers of a coefficient block (cache keys)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.shape, hashlib.blake2b(a.tobytes(), digest_size=8).digest()


def _ids(seq):
    # L matrices are folded into the plan's coefficients: their CONTENT is structure (an
    # in-place edit, or a new array at a recycled address, must not meet a stale plan)
    return tuple(_digest(x) for x in seq) if seq else ()


def _sched(schedule):
    return (schedule.start, schedule.stop, schedule.step) if schedule else None


def structure_fingerprint(costs, limits):
    """Cheap key of everything in the costs / limits that is *structure* for a
    plan (objects, variables, schedules, L contents, field shapes)."""
    key = []
    for name, c in costs.items():
        key.append(("c", name, id(c), c.variable, c.cross, tuple(c.axes), _sched(c.schedule),
                    _ids(c.L), _ids(c.cross_L), np.shape(c.aim), np.shape(c.cross_aim)))
    for l in limits:
        key.append(("l", id(l), l.variable, tuple(l.axes), _sched(l.schedule), _ids(l.L),
                    np.shape(l.arrow), np.shape(l.center), np.shape(l.extreme)))
    return tuple(key)


_EYES = {}


def _eye(n):
    if n not in _EYES:
        _EYES[n] = np.eye(n)
    return _EYES[n]


def formulation_key(form):
    """Everything about a Formulation, besides its costs and limits, that a compiled plan
    depends on: the QP domain, the shapes of the horizon matrices, which coefficient blocks of
    the dynamics' variables are identities, and the CONTENT of every derived definition's
    coefficients (those are folded into the plan's tables; the blocks of the dynamics'
    variables stay outside as rebindable sources).  Two formulations (or two ticks of one)
    with equal keys can share a plan: only sources and parameters differ."""
    key = [tuple(form.domain.items()), tuple(form.optim_variables)]
    for name, dyn in form.dynamics.items():
        key.append((name, type(dyn).__name__,
                    tuple(np.shape(M) for M in getattr(dyn, "matrices", ()))))
    for var, combo in form.definitions.items():
        if var in form.of:
            dyn = form.dynamics[form.of[var]]
            state_ID, domain_ID = getattr(dyn, "state_ID", {}), getattr(dyn, "domain_ID", {})
            items = []
            for dep, m in combo.items():
                m = np.asarray(m, dtype=np.float64)
                # what add_base decides from the numbers: the coefficient IS the slice of the
                # dynamics' horizon matrix (a rebindable source) or a block of its own (a constant
                # source, re-read from the definition) -- a plan compiled for one is wrong for the other
                gather = False
                if var in state_ID and dep in domain_ID:
                    src = np.asarray(dyn.matrices[domain_ID[dep]])
                    sID = state_ID[var]
                    gather = bool(src.ndim == 3 and sID < src.shape[2] and m.shape == src.shape[:2]
                                  and (m == src[..., sID]).all())
                # (asked for only where add_base asks: a slice of the horizon matrix is taken as that)
                eye = (not gather and m.ndim == 2 and m.shape[0] == m.shape[1]
                       and bool((m == _eye(m.shape[0])).all()))
                items.append((dep, m.shape, eye, gather))
            key.append((var, form.of[var], tuple(items)))
        else:
            key.append((var, tuple((dep, _digest(m)) for dep, m in combo.items())))
    return tuple(key)


def _csr_tables(blocks, base_row0, base_rows, total):
    """Stack sparse row blocks and split the global base-row index into
    (base id, row within base)."""
    if blocks:
        M = sp.vstack(blocks, format="csr")
    else:
        M = sp.csr_matrix((0, max(total, 1)))
    starts = np.asarray(base_row0, dtype=np.int64)
    idx = M.indices.astype(np.int64)
    base = np.searchsorted(starts, idx, side="right") - 1 if idx.size else idx
    k = idx - starts[base] if idx.size else idx
    return (M.indptr.astype(np.int32), base.astype(np.int32), k.astype(np.int32),
            M.data.astype(np.float64), M.shape[0])


PM_MAX_OPS = 1 << 20                      # element program of the preview matrices: table limits
PM_MAX_MAP = 1 << 22


def _preview_program(b, rowptr, entbase, entk, entcoef, pmrows):
    """The preview matrices [Mg | Mo] element by element: ``pm_map[r * W + c]`` is the index of
    the element's op list or -1 for a structural zero; element ``i`` is the sum of
    ``pool[cid] * source[sid][offset]`` over ops ``fd_ptr[i] .. fd_ptr[i+1]`` (``sid`` 255: the
    constant 1).  An op is two words: offset inside the source, ``sid | cid << 8``."""
    W = b.ng + b.no
    z = np.zeros(0, dtype=np.int32)
    empty = dict(nfd=0, map=z, fd_ptr=z, ops=z, pool=np.zeros(0))
    if pmrows * W > PM_MAX_MAP or pmrows == 0 or len(b.sources) > 254:
        return empty
    segs_of_base = [[] for _ in b.base_rows]
    for bid, colseg in enumerate(b.colseg):
        for sg in sorted(set(int(x) for x in colseg if x >= 0)):
            segs_of_base[bid].append(b.segments[sg])
    dst, sid_l, off_l, coef = [], [], [], []
    nops = 0
    for r in range(pmrows):
        for e in range(rowptr[r], rowptr[r + 1]):
            u, k, cf = int(entbase[e]), int(entk[e]), float(entcoef[e])
            for sid, off0, rs, es, dst0, length, kind, _ in segs_of_base[u]:
                if kind == SEG_IDENTITY:
                    if k >= length:
                        continue
                    cols = np.array([dst0 + k])
                    sids, offs = np.full(1, 255), np.zeros(1, dtype=np.int64)
                else:
                    cols = dst0 + np.arange(length)
                    sids = np.full(length, sid)
                    offs = off0 + k * rs + np.arange(length, dtype=np.int64) * es
                dst.append(r * W + cols)
                sid_l.append(sids)
                off_l.append(offs)
                coef.append(np.full(cols.size, cf))
                nops += cols.size
        if nops > PM_MAX_OPS:
            return empty
    if not dst:
        return empty
    dst, sids = np.concatenate(dst), np.concatenate(sid_l)
    offs, coef = np.concatenate(off_l), np.concatenate(coef)
    order = np.argsort(dst, kind="stable")          # keeps the entry order inside an element
    dst, sids, offs, coef = dst[order], sids[order], offs[order], coef[order]
    fd_idx, first = np.unique(dst, return_index=True)
    fd_ptr = np.append(first, dst.size).astype(np.int32)
    pool, cid = np.unique(coef, return_inverse=True)
    if pool.size >= 1 << 23 or offs.max(initial=0) >= 1 << 31:
        return empty
    pm_map = np.full(pmrows * W, -1, dtype=np.int32)
    pm_map[fd_idx] = np.arange(fd_idx.size, dtype=np.int32)
    word1 = (sids.astype(np.int64) | (cid.astype(np.int64) << 8)).astype(np.uint32)
    ops = np.stack([offs.astype(np.uint32), word1], axis=1).view(np.int32).reshape(-1)
    return dict(nfd=int(fd_idx.size), map=pm_map, fd_ptr=fd_ptr, ops=ops, pool=pool.astype(np.float64))


def _fused_program(b, rowptr, entbase, entk, entcoef, rtot, ldv):
    """Flatten the row-set program one level further, down to single workspace
    elements: every structurally non-zero element of V (and every d = Mg . given)
    becomes a short list of ops ``coef * arena[src] (* given[g])``, with all
    sources of one instance laid out in one on-chip arena (slot 0 = constant 1)."""
    ng, no = b.ng, b.no
    arena_off, total = [], 1
    for s in b.sources:
        arena_off.append(total)
        total += s.array.size
    arena = np.asarray([[o, s.array.size] for o, s in zip(arena_off, b.sources)],
                       dtype=np.int32).reshape(-1)
    row_tiles = np.zeros(max(rtot, 1), dtype=np.int64)
    empty = dict(ok=0, arena_total=total, arena=arena, fd_idx=np.zeros(0, np.int32),
                 fd_ptr=np.zeros(1, np.int32), ops=np.zeros(0, np.int32),
                 coefpool=np.zeros(0), row_tiles=row_tiles)

    segs_of_base = [[] for _ in b.base_rows]
    for bid, colseg in enumerate(b.colseg):
        for sg in sorted(set(int(x) for x in colseg if x >= 0)):
            segs_of_base[bid].append(b.segments[sg])

    dst, src, gidx, coef = [], [], [], []
    for r in range(rtot):
        for e in range(rowptr[r], rowptr[r + 1]):
            u, k, cf = int(entbase[e]), int(entk[e]), float(entcoef[e])
            for sid, off0, rs, es, dst0, length, kind, _ in segs_of_base[u]:
                if kind == SEG_IDENTITY:
                    if k >= length:
                        continue
                    cols = np.array([dst0 + k])
                    srcs = np.zeros(1, dtype=np.int64)              # the constant 1.0
                else:
                    cols = dst0 + np.arange(length)
                    srcs = arena_off[sid] + off0 + k * rs + np.arange(length) * es
                given = cols < ng
                dst.append(np.where(given, r * ldv + no, r * ldv + (cols - ng)))
                gidx.append(np.where(given, cols, -1))
                src.append(srcs)
                coef.append(np.full(cols.size, cf))
                opt = cols[~given] - ng
                for t in np.unique(opt // 16):
                    row_tiles[r] |= 1 << min(int(t), 30)
        if sum(a.size for a in dst) > FUSED_MAX_OPS:
            empty["row_tiles"] = np.full(max(rtot, 1), 0x7FFFFFFF, dtype=np.int64)
            return empty
    if not dst:
        empty["ok"] = int(total <= FUSED_MAX_ARENA)
        return empty
    dst, src = np.concatenate(dst), np.concatenate(src)
    gidx, coef = np.concatenate(gidx), np.concatenate(coef)
    order = np.argsort(dst, kind="stable")          # keeps the entry order inside an element
    dst, src, gidx, coef = dst[order], src[order], gidx[order], coef[order]
    fd_idx, first = np.unique(dst, return_index=True)
    fd_ptr = np.append(first, dst.size)
    pool, cid = np.unique(coef, return_inverse=True)
    ok = int(total <= FUSED_MAX_ARENA and pool.size < 65536 and ng < 65535)
    packed = (((gidx.astype(np.int64) + 1) << 16) | cid.astype(np.int64)).astype(np.uint32)
    ops = np.stack([src.astype(np.uint32), packed], axis=1).view(np.int32).reshape(-1)
    return dict(ok=ok, arena_total=total, arena=arena, fd_idx=fd_idx.astype(np.int32),
                fd_ptr=fd_ptr.astype(np.int32), ops=ops, coefpool=pool.astype(np.float64),
                row_tiles=row_tiles)


class Workspace:
    """Where the persistent kernel keeps the workspace V in LDS: row major with leading dimension
    ``ldv``; a row holds the columns ``c0[r] .. c0[r] + w[r] - 1`` of the unknowns (its *window*),
    d = Mg.given in column ``vd`` and a one behind it.  Rows come in groups of four (row-sets start
    on a group, zero rows pad them): a 16-row trip feeds lane row lk of the matrix core the rows
    ``base + (0, 8, 4, 12)[lk] + u``, u = k-step -- the two lane rows an 8-byte LDS read serves
    together lie 8 rows apart, which with ldv = 2 (mod 4) is half the banks -- and the two columns
    of a 16-byte piece of G are adjacent.

    *Dense* (:meth:`dense`): every row holds all the unknowns (c0 = 0, vd = no).  *Compact*
    (:meth:`windows`): a row-set holds only the 4-column blocks its rows can be non-zero in -- a
    row-set of an x variable has no y columns -- which is what lets a three-axis problem (C3) fit two
    workgroups per CU.  A trip adds ``- c0 * 8`` to its operand offsets, so that block ``bi`` of the
    unknowns is still found at ``+ 32 bi``; ``row0`` zero rows in front keep those offsets
    non-negative."""

    def __init__(self, no, ldv, vd, row0, c0, w, compact):
        self.no, self.ldv, self.vd, self.row0, self.compact = no, ldv, vd, row0, compact
        self.c0, self.w = np.asarray(c0, dtype=np.int64), np.asarray(w, dtype=np.int64)
        self.rtot = self.c0.size

    @classmethod
    def dense(cls, rtot, no, ldv):
        return cls(no, ldv, no, 0, np.zeros(rtot), np.full(rtot, ldv - 2), 0)

    @classmethod
    def windows(cls, rtot, no, spans, el_row, el_col):
        """Windows from the structural non-zeros ``(el_row, el_col)`` of the workspace, one per
        row-set (``spans``: first row, rows incl. the zero rows behind it)."""
        c0, w = np.zeros(rtot, dtype=np.int64), np.full(rtot, 4, dtype=np.int64)
        keep = el_col < no
        el_row, el_col = el_row[keep], el_col[keep]
        for first, rows in spans:
            cols = el_col[(el_row >= first) & (el_row < first + rows)]
            if cols.size:
                lo, hi = int(cols.min()) // 4 * 4, (int(cols.max()) + 4) // 4 * 4
                c0[first:first + rows], w[first:first + rows] = lo, hi - lo
        ldv = int(w.max()) + 2                                # = 2 mod 4
        first_rows = np.asarray([f for f, _ in spans], dtype=np.int64)
        short = int(max(0, (c0[first_rows] - first_rows * ldv).max())) if len(spans) else 0
        row0 = -(-short // ldv)
        row0 += -row0 % 4
        return cls(no, ldv, int(w.max()), row0, c0, w, 1)

    @property
    def doubles(self):
        return (self.row0 + self.rtot) * self.ldv

    def rowstart(self, r):
        """Index of the first stored element of row r (column c0[r])."""
        return (self.row0 + r) * self.ldv

    def index(self, r, c):
        """Index of element (r, c): c < no an unknown inside the row's window, no: d, no + 1: the one."""
        if c >= self.no:
            return self.rowstart(r) + self.vd + (c - self.no)
        assert self.c0[r] <= c < self.c0[r] + self.w[r], "outside the row's window"
        return self.rowstart(r) + c - int(self.c0[r])

    def origin(self, r):
        """What a trip adds to 8 * 4 * block to find the block in row r: bytes."""
        return (self.rowstart(r) - int(self.c0[r])) * 8

    def holds_block(self, r, blk):
        return self.c0[r] <= 4 * blk and 4 * blk + 4 <= self.c0[r] + self.w[r]


def _resident_rows(limit_recs, lax_recs, nparams, ws):
    """Per row of the stacked G, 16 words: workspace index (:meth:`Workspace.rowstart`) of every axis' row [4], arrow
    param of every axis [4], center param of every axis [4], naxes, extreme param, then
    the first two axes once more, packed: voff0 | voff1 << 16, arrow0 | arrow1 << 16 (zero
    where that does not fit); a missing axis points at workspace row 0 with the always-zero
    parameter slot ``nparams``."""
    rows = []
    for out0, nrows, naxes, lax0, p_a, a_rows, p_c, c_rows, p_e, e_rows, _, _ in limit_recs:
        for r in range(nrows):
            tail = [naxes, p_e + (0 if e_rows == 1 else r), 0, 0]
            voff, ap, cp = [], [], []
            for ax in range(RS_AXMAX):
                if ax < naxes:
                    off, rs = lax_recs[lax0 + ax]
                    voff.append(ws.rowstart(off + (0 if rs == 1 else r)))
                    ap.append(p_a + (0 if a_rows == 1 else r) * naxes + ax)
                    cp.append(p_c + (0 if c_rows == 1 else r) * naxes + ax)
                else:
                    voff.append(0)
                    ap.append(nparams)
                    cp.append(nparams)
            if naxes <= 2 and max(voff) < 65536 and nparams < 65536:
                tail[2:] = [voff[0] | (voff[1] << 16), ap[0] | (ap[1] << 16)]
            rows.append(voff + ap + cp + tail)
    return np.asarray(rows, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1)


def _lti_groups(form, sources, names):
    """Source groups whose horizon matrices the persistent kernel generates on chip from
    the system's own ``(A, B)``: for every dynamics name in ``names`` the sources
    ``(name, 0..m-1)`` = ``U_j`` ``(N, N, n)`` and ``(name, m)`` = ``S`` ``(N, n, n)``
    (``matrices = U + [S]``, dynamics.py:199; S[k,j,i] = (A^{k+1})[i,j],
    U_j[k,l,i] = (A^{k-l} B)[i,j], tools.py:14-33)."""
    groups = []
    for name in names:
        dyn = form.dynamics[name]
        m = len(dyn.matrices) - 1
        ids = {s.key[1]: i for i, s in enumerate(sources) if s.key[0] == name and len(s.key) == 2}
        if m < 1 or sorted(ids) != list(range(m + 1)):
            raise ValueError("dynamics %r: every horizon matrix U_0..U_%d, S must be used by the "
                             "formulation to generate them on chip" % (name, m - 1))
        N, n = sources[ids[m]].array.shape[0], sources[ids[m]].array.shape[1]
        if sources[ids[m]].array.shape != (N, n, n) or any(
                sources[ids[j]].array.shape != (N, N, n) for j in range(m)):
            raise ValueError("dynamics %r: unexpected shapes of the horizon matrices" % name)
        groups.append(dict(name=name, n=n, m=m, N=N, ids=[ids[k] for k in range(m + 1)]))
    if len(groups) > RS_LTI_MAX:
        raise ValueError("at most %d source groups can be generated on chip" % RS_LTI_MAX)
    return groups


def _resident_image(sources, ng, nparams, groups=()):
    """The input image of one instance in LDS and the LDS-DMA loads that fill it.

    Doubles, in order: ``1, 1 | source 0 | source 1 ... | given, 1 | params, 0 | 0 ...``
    (a 1.0 behind ``given`` for ops without a given factor, a 0.0 behind ``params`` for the
    missing axes of a constraint row; every block starts on an even offset; this part, which
    the loads fill, is a multiple of 128 doubles).  The sources of a group that is
    generated on chip are not loaded: the group's ``A`` (stream of its first source) and
    ``B`` (stream of its second) are, and behind the loaded part the image has room for
    the tables ``(A^{k+1})[i][j]``, ``(A^d B)[i][j]`` (k, d < N) and for the powers
    ``A^(2^s)`` they are built from.  Input streams: the sources, then given, params, and
    a constant stream ``[1, 1, 0, 0]`` owned by the plan.  One load moves ``unit`` bytes
    per lane, 64 lanes to consecutive LDS addresses: 16 when every pair of doubles of the
    image comes from consecutive, even-aligned elements of one stream, else 4.
    Returns the per-lane table ``meta[nchunk * 64][2]`` = (stream, byte offset)."""
    nsrc = len(sources)
    s_given, s_params, s_const = nsrc, nsrc + 1, nsrc + 2
    slots = []                                   # per double of the image: (stream, element)
    generated = {i: g for g in groups for i in g["ids"]}

    def push_const(first):                       # 1.0 is element 0 / 1, 0.0 is 2 / 3
        slots.append((s_const, first))
        if len(slots) & 1:
            slots.append((s_const, first + 1))

    def pad_even():
        if len(slots) & 1:
            slots.append((s_const, 3))

    push_const(0)
    src_off = []
    for sid, src in enumerate(sources):
        src_off.append(len(slots))
        if sid in generated:
            continue
        slots.extend((sid, k) for k in range(src.array.size))
        pad_even()
    given_off = len(slots)
    slots.extend((s_given, k) for k in range(ng))
    push_const(0)
    pad_even()
    params_off = len(slots)
    slots.extend((s_params, k) for k in range(nparams))
    push_const(2)
    pad_even()
    while len(slots) % 128:
        slots.extend([(s_const, 2), (s_const, 3)])
    arr = np.asarray(slots, dtype=np.int64)
    even, odd = arr[0::2], arr[1::2]
    paired = bool(np.all((even[:, 0] == odd[:, 0]) & (odd[:, 1] == even[:, 1] + 1)
                         & (even[:, 1] % 2 == 0)))
    if paired:
        unit = 16
        meta = np.stack([even[:, 0], even[:, 1] * 8], axis=1)
    else:
        unit = 4
        meta = np.stack([np.repeat(arr[:, 0], 2),
                         (np.repeat(arr[:, 1] * 8, 2) + np.tile([0, 4], len(arr)))], axis=1)
    # The (A, B) of the generated groups travel apart from the image, two instances ahead,
    # into a ring of two slots (4-byte loads: A has n^2 doubles, no pairing to rely on): the
    # tables of instance i+1 are built while instance i is assembled.
    ab = []
    for g in groups:
        g["img_a"] = len(ab)                     # A through the group's first stream ...
        ab.extend((g["ids"][0], k) for k in range(g["n"] * g["n"]))
        ab.extend([(s_const, 2)] * (len(ab) & 1))
        g["img_b"] = len(ab)                     # ... B through its second
        ab.extend((g["ids"][1], k) for k in range(g["n"] * g["m"]))
        ab.extend([(s_const, 2)] * (len(ab) & 1))
    while len(ab) % 32:
        ab.append((s_const, 2))
    ab_arr = np.asarray(ab, dtype=np.int64).reshape(-1, 2)
    ab_meta = np.stack([np.repeat(ab_arr[:, 0], 2),
                        np.repeat(ab_arr[:, 1] * 8, 2) + np.tile([0, 4], len(ab_arr))], axis=1)
    total = len(slots)
    for g in groups:                             # tables behind the loaded part
        n, m, N = g["n"], g["m"], g["N"]
        g["stages"] = max(1, int(np.ceil(np.log2(N)))) if N > 1 else 1
        g["tab_a"], total = total, total + N * n * n
        g["tab_b"], total = total, total + N * n * m
        g["tab_p"], total = total, total + (g["stages"] + 1) * n * n
        total += total & 1
    return dict(unit=unit, nchunk=meta.shape[0] // 64, meta=meta.astype(np.int32).reshape(-1),
                img=total, dma=len(slots), given=given_off, params=params_off, src_off=src_off,
                groups=list(groups), ab=len(ab), ab_meta=ab_meta.astype(np.int32).reshape(-1))


def _lti_table_offset(g, k, flat):
    """Image offset of element ``flat`` of the group's k-th horizon matrix inside the
    generated tables, or None where the matrix is structurally zero (U above the diagonal)."""
    n, m, N = g["n"], g["m"], g["N"]
    if k == m:                                   # S[kk][j][i] = (A^{kk+1})[i][j]
        kk, rem = divmod(flat, n * n)
        j, i = divmod(rem, n)
        return g["tab_a"] + (kk * n + i) * n + j
    kk, rem = divmod(flat, N * n)                # U_k[kk][l][i] = (A^{kk-l} B)[i][k]
    l, i = divmod(rem, n)
    if l > kk:
        return None
    return g["tab_b"] + ((kk - l) * n + i) * m + k


def _resident_program(fused, gterms, no, ldv, ws, image, ng, nparams, nc_rows):
    """Tables of the persistent fused kernel: the compose ops of ``_fused_program``
    dealt out to the RS_NT threads of a workgroup (kept in registers for the whole
    launch) and the Hessian + gradient work split into per-wavefront lists of MFMA items.
    ``ldv``: the leading dimension ``fused["fd_idx"]`` is written in; ``ws``: where the kernel
    keeps the element (:class:`Workspace`)."""
    import heapq

    NT, NW = RS_NT, RS_NW
    z = np.zeros(0, dtype=np.int32)
    out = dict(ok=0, jc=0, sym=0, src=z, gidx=z, dst=z, coef=np.zeros(0), trips=z, ntrip=0, split=z, zblk=z,
               wtrip=np.zeros(RS_WAVES * 2, dtype=np.int32))
    if not fused["ok"] or image["img"] > 65535 or nparams > 65535:
        return out
    on_column = np.zeros(no + 1, dtype=np.int64)
    for g in gterms:
        if g[6] & GT_FLAG_DIAG:
            on_column[g[0]:g[0] + g[2]] += 1
    if on_column.max() > RS_DIAG_MAX:
        return out
    # ---- compose: elements to threads, longest first onto the lightest thread; an element
    # with more ops than a thread may hold is shared by two threads (both add their part)
    fd_idx, fd_ptr = fused["fd_idx"], fused["fd_ptr"]
    ops = fused["ops"].view(np.uint32).reshape(-1, 2)
    pool = fused["coefpool"]
    arena = fused["arena"].reshape(-1, 2)
    generated = {i: (g, k) for g in image["groups"] for k, i in enumerate(g["ids"])}

    def image_offset(a):                          # fused arena offset -> image offset, or None
        if a == 0:
            return 0                              # the constant 1.0
        for sid, (off, size) in enumerate(arena):
            if off <= a < off + size:
                if sid in generated:
                    return _lti_table_offset(*generated[sid], a - off)
                return image["src_off"][sid] + a - off
        raise AssertionError("arena offset outside every source")

    # ops that read a structural zero of a generated source are dropped (exact zeros)
    op_img = [image_offset(int(a)) for a in ops[:, 0]]
    kept = [[o for o in range(int(fd_ptr[i]), int(fd_ptr[i + 1])) if op_img[o] is not None]
            for i in range(len(fd_idx))]
    counts = np.asarray([len(k) for k in kept], dtype=np.int64)
    owner, split = None, []
    for cap in (3, 5, 8, RS_JC_MAX):
        if counts.size and counts.max() > 2 * cap:
            continue
        pieces = []                               # (first, last position in kept[i], element, shared)
        for i, c in enumerate(counts):
            if c == 0:
                continue                          # the element stays zero
            if c > cap:
                mid = (int(c) + 1) // 2
                pieces += [(0, mid, i, True), (mid, int(c), i, True)]
            else:
                pieces.append((0, int(c), i, False))
        heap = [(0, t) for t in range(NT)]
        heapq.heapify(heap)
        trial = [[] for _ in range(NT)]
        for pc in sorted(pieces, key=lambda pc: (pc[0] - pc[1], pc[2], pc[0])):
            load, t = heapq.heappop(heap)
            trial[t].append(pc)
            heapq.heappush(heap, (load + pc[1] - pc[0], t))
        if max((sum(pc[1] - pc[0] for pc in own) for own in trial), default=0) <= cap:
            owner = trial
            split = sorted({ws.index(*divmod(int(fd_idx[pc[2]]), ldv)) for pc in pieces if pc[3]})
            break
    if owner is None:
        return out
    jc = max(1, max((sum(pc[1] - pc[0] for pc in own) for own in owner), default=0))
    src = np.zeros((jc, NT), dtype=np.int32)            # image[0] = 1.0
    gidx = np.full((jc, NT), image["given"] + ng, dtype=np.int32)   # ... and given[ng] = 1.0
    dst = -np.ones((jc, NT), dtype=np.int32)
    coef = np.zeros((jc, NT))
    for t, own in enumerate(owner):
        j = 0
        for lo, hi, i, shared in sorted(own, key=lambda pc: (pc[2], pc[0])):
            for o in kept[i][lo:hi]:
                src[j, t] = op_img[o]
                gi = (int(ops[o, 1]) >> 16) - 1
                if gi >= 0:
                    gidx[j, t] = image["given"] + gi
                coef[j, t] = pool[int(ops[o, 1]) & 0xFFFF]
                j += 1
            dst[j - 1, t] = ws.index(*divmod(int(fd_idx[i]), ldv)) | (RS_DST_ACC if shared else 0)
    # ---- Hessian and gradient on the matrix core, in 4x4 blocks (plan_tables.h RT_*).
    # Block (bi, bj) of P exists when some term has structural non-zeros in columns 4bi.. of
    # its A rows and 4bj.. of its B rows; block bi of q when a term's A rows reach columns
    # 4bi...  Diagonal blocks and all blocks of q always exist (the diagonal gterms add into
    # them).  Four blocks with (nearly) the same set of terms share a pack = an accumulator;
    # a *trip* is up to 16 rows of one term into one pack.
    nb = (no + 3) // 4
    if nb > RS_BLOCKS_MAX:
        return out
    terms = [g for g in gterms if not g[6] & GT_FLAG_DIAG]
    pterms = [g for g in terms if g[6] & GT_FLAG_P]
    sym = int(all(g[0] == g[1] for g in pterms))
    rtot_rows = max([g[0] + g[2] for g in terms] + [max(g[1], g[4]) + g[2] for g in terms] + [1])
    reach = np.zeros((rtot_rows, nb), dtype=bool)    # structural non-zeros of V by block column
    el_row, el_col = fd_idx // ldv, fd_idx % ldv
    keep = (el_col < no) & (el_row < rtot_rows) & (counts > 0)   # (no kept op: an exact zero)
    reach[el_row[keep], el_col[keep] // 4] = True
    p_blocks = {(bi, bi): [] for bi in range(nb)}
    q_blocks = {bi: [] for bi in range(nb)}
    for gi, g in enumerate(terms):
        aoff, boff, nrows, flags = g[0], g[1], g[2], g[6]
        ablk = np.flatnonzero(reach[aoff:aoff + nrows].any(axis=0))
        for bi in ablk:
            q_blocks[int(bi)].append(gi)
        if flags & GT_FLAG_P:
            bblk = np.flatnonzero(reach[boff:boff + nrows].any(axis=0))
            for bi in ablk:
                for bj in bblk:
                    if not sym or bi <= bj:
                        p_blocks.setdefault((int(bi), int(bj)), []).append(gi)
    # Blocks with the same set of terms fill packs of four; what is left over is merged
    # where it costs the fewest extra trips (a pack is visited by every term of its blocks).
    # A block of q is (bi, -1).
    blocks = dict(p_blocks)
    blocks.update({(bi, -1): v for bi, v in q_blocks.items()})
    by_sig = {}
    for key in sorted(blocks):
        by_sig.setdefault(tuple(blocks[key]), []).append(key)
    packs, partial = [], []                          # ([block] * <= 4, set of term ids)
    for sig in sorted(by_sig):
        keys = by_sig[sig]
        for i in range(0, len(keys), 4):
            (packs if len(keys) - i >= 4 else partial).append((keys[i:i + 4], set(sig)))
    def in_window(key, gi):
        """Block ``key`` inside the windows of term gi's rows: what the term reads there are true
        elements of the workspace (zeros where it does not reach), not some other row's."""
        aoff, boff, flags = terms[gi][0], terms[gi][1], terms[gi][6]
        if not ws.holds_block(aoff, key[0]):
            return False
        return key[1] < 0 or ws.holds_block(boff if flags & GT_FLAG_P else aoff, key[1])

    merged = []
    for keys, sig in sorted(partial, key=lambda p: (-len(p[0]), sorted(p[1]))):
        best = None
        for m in merged:
            if len(m[0]) + len(keys) <= 4 and (not ws.compact or (
                    all(in_window(k, gi) for k in keys for gi in m[1] - sig)
                    and all(in_window(k, gi) for k in m[0] for gi in sig - m[1]))):
                extra = len(m[1] | sig) * 2 - len(m[1]) - len(sig)
                if best is None or extra < best[0]:
                    best = (extra, m)
        if best is None:
            merged.append((list(keys), set(sig)))
        else:
            best[1][0].extend(keys)
            best[1][1].update(sig)
    packs += merged
    pack_trips = []
    for grp, tids in packs:
        live = (1 << len(grp)) - 1
        qmask = sum(1 << j for j, k in enumerate(grp) if k[1] < 0)
        grp = [(k[0], max(k[1], 0)) for k in grp]
        grp = grp + [grp[0]] * (4 - len(grp))
        bi_bytes = sum(k[0] << (8 * j) for j, k in enumerate(grp))
        bj_bytes = sum(k[1] << (8 * j) for j, k in enumerate(grp))
        lst = []
        row_bytes = ws.ldv * 8
        for gi in sorted(tids):
            aoff, boff, nrows, wparam, doff, aimparam, flags = terms[gi][:7]
            half = 1 if flags & GT_FLAG_HALF else 0
            nop = 0 if flags & GT_FLAG_P else 1
            brow = aoff if nop else boff                 # (a term without P only meets q lanes)
            assert aoff % 4 == 0 and brow % 4 == 0 and doff % 4 == 0
            n4 = nrows + (-nrows % 4)                    # the zero rows behind the row-set included
            k0, mine = 0, []
            while k0 < n4:
                # full trips of 16 rows (four groups, four k-steps); what is left goes group by
                # group: *short* trips, one k-step each
                rows = 16 if n4 - k0 >= 16 else 4
                mine.append([ws.origin(aoff) + k0 * row_bytes, ws.origin(brow) + k0 * row_bytes,
                             (ws.rowstart(doff + k0) + ws.vd) * 8, wparam * 8, aimparam * 8,
                             rows | ((rows == 4) << RT_SHORT) | (half << RT_HALF) | (nop << RT_NOP),
                             bi_bytes, bj_bytes])
                k0 += rows
            mine[-1][RT_WORD] |= 1 << RT_TERM_END        # the term's sum gets its weight here
            lst += mine
        if not lst:                                  # nothing to add up: the pack is still written
            lst.append([0, 0, 0, 0, 0, 0, bi_bytes, bj_bytes])
        for trip in lst:
            trip[RT_WORD] |= (live << RT_LIVE) | (qmask << RT_QMASK)
        lst[0][RT_WORD] |= 1 << RT_FIRST
        lst[-1][RT_WORD] |= 1 << RT_LAST
        pack_trips.append(lst)
    # packs to wavefronts, heaviest first onto the least loaded.  Costs in (measured,
    # rounded) cycles of one wavefront: a trip, a pack store, a 16-byte piece of G, a
    # chunk of the input fetch.  The matrix waves start with the fetch on their account,
    # the stream waves with their share of G.
    def trip_cost(trip):
        word = trip[RT_WORD]
        if not word & 31:
            return 0
        return (TRIP_COST + (TRIP_STEP_COST if (word >> RT_SHORT) & 1 else 4 * TRIP_STEP_COST)
                + (TERM_COST + (TRIP_Q_COST if (word >> RT_QMASK) & 15 else 0)
                   if (word >> RT_TERM_END) & 1 else 0))

    cost = [PACK_COST + sum(trip_cost(trip) for trip in lst) for lst in pack_trips]
    stream_threads = NT - NW * 64
    pieces = nc_rows * max(no // 2, 1)
    by_descriptor = no % 2 == 0 and pieces <= RS_GDESC_PIECES * RS_GDESC_THREADS   # (compile_plan: rs_gdesc)

    loads = []
    for w in range(RS_WAVES):
        if w < NW:
            gen = TABLES_COST if image["groups"] and w == NW - 1 else 0   # builds the horizon tables
            loads.append((FETCH_CHUNK_COST * len(range(w, image["nchunk"], NW)) + gen, w))
        else:                                      # threads of this wave own pieces e = wt + u WT
            first = (w - NW) * 64
            own = len(range(first, pieces, stream_threads))
            loads.append(((G_DESC_PIECE_COST if by_descriptor else G_PIECE_COST) * own + STREAM_WAVE_COST, w))
    heapq.heapify(loads)
    wave_packs = [[] for _ in range(RS_WAVES)]
    for k in sorted(range(len(pack_trips)), key=lambda k: -cost[k]):
        load, w = heapq.heappop(loads)
        wave_packs[w].append(k)
        heapq.heappush(loads, (load + cost[k], w))
    trips, wtrip = [], np.zeros(RS_WAVES * 2, dtype=np.int32)
    for w in range(RS_WAVES):
        wtrip[2 * w] = len(trips)
        for k in wave_packs[w]:
            trips.extend(pack_trips[k])
        wtrip[2 * w + 1] = len(trips) - wtrip[2 * w]
    ntrip = len(trips)
    trips += [[0] * RS_TRIP_WORDS] * 2             # the kernel reads two records ahead
    # blocks of P no term reaches (both triangles): exact zeros; the kernel that keeps P in LDS
    # zeroes them once per workgroup, the one that writes P straight to HBM (wide problems)
    # writes them per instance
    reached = set()
    for (bi, bj) in p_blocks:
        reached.add((bi, bj))
        if sym:
            reached.add((bj, bi))
    zblk = np.asarray([(bi << 8) | bj for bi in range(nb) for bj in range(nb)
                       if (bi, bj) not in reached], dtype=np.int32)
    out.update(zblk=zblk)
    out.update(ok=1, jc=jc, sym=sym, src=src.reshape(-1), gidx=gidx.reshape(-1),
               dst=dst.reshape(-1), coef=coef.reshape(-1),
               trips=np.asarray(trips, dtype=np.int32).reshape(-1), ntrip=ntrip, wtrip=wtrip,
               split=np.asarray(split, dtype=np.int32))
    return out


def is_causal(U):
    """Whether a ``(..., N, N, n)`` array holds exact zeros above the diagonal: ``U[.., k, l, :] == 0``
    for ``l > k`` -- what ``tools.extend_matrices`` (tools.py:27-31) and ``mpcasm_fill_su`` produce."""
    U = np.asarray(U)
    N = U.shape[-3]
    kk, ll = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    return not np.any(U[..., ll > kk, :])


def _packed_program(resident):
    """The persistent kernel's compose program once more, as ONE 16-byte record per op and thread --
    image offset of the source | of the given value << 16, destination, the coefficient's two halves --
    so that a thread's set-up is one load per op instead of four (plan_tables.h H_OFF_RS_PROG)."""
    jc = int(resident.get("jc", 0)) if resident.get("ok") else 0
    if jc == 0:
        return np.zeros(0, dtype=np.int32)
    src = np.asarray(resident["src"], dtype=np.int64).reshape(jc, RS_NT)
    gidx = np.asarray(resident["gidx"], dtype=np.int64).reshape(jc, RS_NT)
    dst = np.asarray(resident["dst"], dtype=np.int64).reshape(jc, RS_NT)
    coef = np.ascontiguousarray(np.asarray(resident["coef"], dtype=np.float64).reshape(jc, RS_NT))
    rec = np.zeros((jc, RS_NT, 4), dtype=np.int32)
    rec[:, :, 0] = (src | (gidx << 16)).astype(np.uint32).view(np.int32)
    rec[:, :, 1] = dst.astype(np.int32)
    rec[:, :, 2:4] = coef.view(np.int32).reshape(jc, RS_NT, 2)
    return rec.reshape(-1)


def causal_sources(form, sources, groups=()):
    """Ids of the sources that are the ``U_j`` of an ExtendedSystem and hold exact zeros above
    the diagonal: ``U_j[k][l] = A^(k-l) B`` for ``l <= k``, else 0 (tools.py:27-31).  The tables
    a plan derives from this (tile masks, CSC patterns) hold for every instance as long as the
    tensors bound per instance are causal as well -- what ``tools.extend_matrices`` and
    ``mpcasm_fill_su`` produce.  Sources generated on chip are causal by construction."""
    out = {i for g in groups for i in g["ids"][:-1]}
    for sid, src in enumerate(sources):
        key = src.key
        if sid in out or len(key) != 2 or key[0] not in form.dynamics or src.array.ndim != 3:
            continue
        m = len(getattr(form.dynamics[key[0]], "matrices", [])) - 1
        N = src.array.shape[0]
        if key[1] < m and src.array.shape[1] == N and is_causal(src.array):
            out.add(sid)
    return out


def _segments_of_base(b):
    out = [[] for _ in b.base_rows]
    for bid, colseg in enumerate(b.colseg):
        for sg in sorted(set(int(x) for x in colseg if x >= 0)):
            out[bid].append(b.segments[sg])
    return out


def _column_tables(b, groups, nop):
    """``ci[base][column][2]`` over the columns ``[given | unknowns padded to nop]``
    (csrc/plan_tables.h, H_T_CI_OK): element offset for base row 0 and ``rs | stream << 24``.
    Offsets into the stream that is the plan's dtab are relative to the delta table here
    (``compile_plan`` adds its position).  Also returns the delta table and ok."""
    ng, no = b.ng, b.no
    nbase = len(b.base_rows)
    L = max([seg[5] for seg in b.segments if seg[6] == SEG_IDENTITY] + [0])
    delta = np.zeros(2 * L + 2)
    delta[1 + L] = 1.0                                  # delta[1 + L + j - k] = (j == k)
    ci = np.zeros((nbase, ng + nop, 2), dtype=np.int64)
    ci[:, :, 1] = T_SID_CONST << 24                     # no segment: the 0.0 at delta[0], rs = 0
    gen = {}
    for g in groups:
        for j, sid in enumerate(g["ids"][:-1]):
            gen[sid] = (g, j)
    ok = True
    for bid, segs in enumerate(_segments_of_base(b)):
        for sid, off0, rs, es, dst0, length, kind, _ in segs:
            j = np.arange(length, dtype=np.int64)
            cols = dst0 + j                              # (a domain variable is all given or all optim)
            if kind == SEG_IDENTITY:
                off, stream, step = 1 + L + j, T_SID_CONST, -1
            elif sid in gen:
                # U_j[k][l][i] = TB[i][j][N + k - l] (plan_tables.h TL_*): offset for k = 0 and one
                # element per base row
                g, jj = gen[sid]
                n, m, N = g["n"], g["m"], g["N"]
                if rs != N * n or es != n or length != N or not 0 <= off0 < n:
                    ok = False
                    continue
                off, stream, step = (off0 * m + jj) * 2 * N + N - j, sid, 1
            else:
                off, stream, step = off0 + j * es, sid, rs
            if abs(step) >= 1 << 23 or off.max(initial=0) >= 1 << 31 or off.min(initial=0) < 0:
                ok = False
                continue
            ci[bid, cols, 0] = off
            ci[bid, cols, 1] = (step & 0xFFFFFF) | (stream << 24)
    return ci, delta, ok


def _row_tile_masks(b, rowptr, entbase, entk, rtot, causal):
    """Per workspace row: bit t set when 16-column tile t of the unknowns can hold a non-zero
    (tiles beyond 62 fold onto bit 63).  A causal source contributes columns l <= k only."""
    ng = b.ng
    segs_of_base = _segments_of_base(b)
    masks = [0] * max(rtot, 1)
    for r in range(rtot):
        m = 0
        for e in range(rowptr[r], rowptr[r + 1]):
            u, k = int(entbase[e]), int(entk[e])
            for sid, off0, rs, es, dst0, length, kind, _ in segs_of_base[u]:
                if dst0 < ng or length == 0:
                    continue
                c0 = dst0 - ng
                if kind == SEG_IDENTITY:
                    if k >= length:
                        continue
                    lo = hi = c0 + k
                elif sid in causal:
                    lo, hi = c0, c0 + min(k, length - 1)
                else:
                    lo, hi = c0, c0 + length - 1
                for t in range(lo // 16, hi // 16 + 1):
                    m |= 1 << min(t, 63)
        masks[r] = m
    return masks


def _tiled_program(b, form, gterms, rowptr, entbase, entk, entcoef, rtot, groups, rr_ok, g_rows):
    """Tables of the tiled kernel (csrc/tiled.hip): the column tables, the stage list of the
    Hessian / gradient terms with their structural tile masks, and the groups of horizon
    tables a pre-pass generates from per-instance (A, B)."""
    ng, no = b.ng, b.no
    nop = -(-max(no, 1) // T_BLOCK) * T_BLOCK
    ci, delta, ci_ok = _column_tables(b, groups, nop)
    causal = causal_sources(form, b.sources, groups)
    masks = _row_tile_masks(b, rowptr, entbase, entk, rtot, causal)
    nent_of = np.diff(rowptr) if rtot else np.zeros(0, dtype=np.int64)

    def simple(row0, n):
        """The common base of rows that are one entry each, else -1."""
        if n == 0 or np.any(nent_of[row0:row0 + n] != 1):
            return -1
        bases = entbase[rowptr[row0:row0 + n]]
        return int(bases[0]) if np.all(bases == bases[0]) else -1

    def ormask(row0, n):
        m = 0
        for r in range(row0, row0 + n):
            m |= masks[r]
        return m

    def quarter_class(m):
        """Smallest n such that in every 64-column quarter only the first n tiles are set."""
        n = 0
        while m:
            n = max(n, (m & 15).bit_length())
            m >>= 4
        return n

    # rows of G that are arrow * (one workspace row): they ride on the first stage whose A rows
    # hold that row (g_rows: per row of G the workspace rows of its axes)
    riders = {}
    for R, axes in enumerate(g_rows):
        if len(axes) == 1:
            riders.setdefault(axes[0][0], []).append((R, axes[0][1]))
    stages, srow, scoef, pig = [], [], [], []
    for g in gterms:
        aoff, boff, nrows, wparam, doff, aimparam, flags = g[:7]
        if flags & GT_FLAG_DIAG:
            continue
        has_p = bool(flags & GT_FLAG_P)
        for k0 in range(0, nrows, 16):
            n = min(16, nrows - k0)
            arow, drow = aoff + k0, doff + k0
            brow = boff + k0 if has_p else arow
            fl = (TS_FLAG_P if has_p else 0) | (TS_FLAG_HALF if flags & GT_FLAG_HALF else 0)
            ba, bb = simple(arow, n), simple(brow, n)
            if 0 <= ba < 65536:
                fl |= TS_FLAG_SIMPLE_A
            if 0 <= bb < 65536:
                fl |= TS_FLAG_SIMPLE_B
            if brow == arow:
                fl |= TS_FLAG_SAME
            ma, mb = ormask(arow, n), ormask(brow, n)
            cls = max(1, quarter_class(ma), quarter_class(mb)) if has_p else 0
            ks, cs = np.zeros((2, 16), dtype=np.int64), np.zeros((2, 16))
            for side, (row0, smp) in enumerate(((arow, ba), (brow, bb))):
                if 0 <= smp < 65536:
                    ks[side, :n] = entk[rowptr[row0:row0 + n]]
                    cs[side, :n] = entcoef[rowptr[row0:row0 + n]]
            pg = -np.ones((16, T_PIG_MAX, 2), dtype=np.int64)
            for i in range(n):
                take = riders.get(arow + i, [])[:T_PIG_MAX]
                for t, (R, slot) in enumerate(take):
                    pg[i, t] = (R, slot)
                    fl |= TS_FLAG_G
                if take:
                    riders[arow + i] = riders[arow + i][len(take):]
            # a window of the group's Toeplitz table: rows k0 .. k0 + n - 1 of one generated state,
            # one coefficient (plan_tables.h TS_FLAG_TOEPLITZ)
            ua = ub = sba = sbb = 0
            if len(groups) == 1:
                g0 = groups[0]
                state_of = {bid: dyn_state for bid, dyn_state in b.lti_state_of(g0).items()}

                def window(side, smp):
                    if not 0 <= smp < 65536 or smp not in state_of:
                        return None
                    k, c = ks[side, :n], cs[side, :n]
                    if np.any(k != k[0] + np.arange(n)) or np.any(c != c[0]):
                        return None
                    sboff = state_of[smp] * g0["m"] * 2 * g0["N"]
                    return sboff + int(k[0]), sboff

                wa = window(0, ba)
                wb = window(1, bb) if has_p else wa
                if wa is not None and wb is not None:
                    fl |= TS_FLAG_TOEPLITZ
                    (ua, sba), (ub, sbb) = wa, wb
            stages.append([arow, brow, drow, n | (fl << 8) | (cls << 16), wparam, aimparam,
                           ma & 0xFFFFFFFF, ma >> 32, mb & 0xFFFFFFFF, mb >> 32,
                           (max(ba, 0) & 0xFFFF) | ((max(bb, 0) & 0xFFFF) << 16), 0,
                           ua, ub, sba, sbb])
            srow.append(ks)
            scoef.append(cs)
            pig.append(pg)
    order = sorted(range(len(stages)), key=lambda i: stages[i][3] >> 16)     # by class, stable
    stages = [stages[i] for i in order]
    srow, scoef, pig = ([x[i] for i in order] for x in (srow, scoef, pig))
    riding = {int(R) for pg in pig for R in pg[:, :, 0].ravel() if R >= 0}
    grest = np.asarray([R for R in range(len(g_rows)) if R not in riding], dtype=np.int64)
    d_len = rtot + (rtot & 1)
    work = d_len
    lti, ids = [], []
    for g in groups:
        n, m, N = g["n"], g["m"], g["N"]
        lti.append([n, m, N, len(ids), work, work + N * n * n, 0, 0])
        work += N * n * n + n * m * 2 * N
        ids.extend(g["ids"])
    ok = int(ci_ok and rr_ok and no >= T_BLOCK and len(b.base_rows) > 0
             and len(b.sources) <= MAX_SOURCES and rtot < (1 << 24))
    toeplitz = int(bool(ok and stages and all((st[3] >> 8) & TS_FLAG_TOEPLITZ for st in stages)))
    return dict(ok=ok, toeplitz=toeplitz, ci_ok=int(ci_ok), nop=nop, ci=ci, delta=delta, masks=masks,
                causal=causal,
                srow=np.asarray(srow, dtype=np.int64).reshape(-1),
                scoef=np.asarray(scoef, dtype=np.float64).reshape(-1),
                pig=np.asarray(pig, dtype=np.int64).reshape(-1), grest=grest,
                stages=np.asarray(stages, dtype=np.int64).reshape(-1, T_STAGE_WORDS),
                lti=np.asarray(lti, dtype=np.int64).reshape(-1, T_LTI_WORDS),
                lti_ids=np.asarray(ids, dtype=np.int64), work=work)


def _scan_tables(b, gterms, rowptr, entbase, entk, entcoef, groups, g_rows, tiled, nparams):
    """Tables of the tiled kernel's *scan form* (csrc/tiled.hip ``toeplitz_scan_kernel``,
    plan_tables.h T_SCAN*).

    When every Hessian term is ``w (c M_i)^T (c M_i)`` over ALL N rows of one state ``i`` of the
    plan's single generated LTI group, ``M_i[k][(j, l)] = T_ij[k - l]`` (``T_ij[d] = (A^d B)[i][j]``
    for ``d >= 0``, else 0; tools.py:27-31) and the block of ``P`` on the columns of inputs j, j'
    obeys ``P[(j,l)][(j',l')] = C[(j,l)][(j',l')] + P[(j,l+1)][(j',l'+1)]`` with the rank-K term
    ``C[r][c] = sum_g w_g c_g^2 M_g[N-1][r] M_g[N-1][c]`` (the last row of every state) and nothing
    behind l = N-1: the Hessian is a sum along diagonals, O(K) multiply-adds per element instead
    of the O(K N) of the product.  The kernel keeps the table without its zero halves,
    ``Tc[(i m + j) N + d] = T_ij[d]``.  The tables: the column blocks ``(first column, j N)`` of the
    inputs that are unknowns, the K terms ``(i m N, weight slot, aim slot, first row of d)`` with their
    coefficients, and per row of G ``(i m N + k, arrow slot)`` with its coefficient when the row is
    ``arrow * c * (row k of state i)`` (else -1: the row is composed through the column tables behind
    the rest)."""
    nc, no, ng = len(g_rows), b.no, b.ng
    off = dict(ok=0, K=0, blk=np.zeros((0, 2), np.int64), gt=np.zeros((0, T_SCAN_GT_WORDS), np.int64),
               gc=np.zeros(0), grow=np.zeros((0, 2), np.int64), gcoef=np.zeros(0),
               grest=np.zeros(0, np.int64), colblk=np.zeros(0, np.int64), nother=0, fused=0)
    if not tiled["toeplitz"] or len(groups) != 1:
        return off
    g0 = groups[0]
    n, m, N = g0["n"], g0["m"], g0["N"]
    if N > T_SCAN_NMAX or nparams >= 1 << 20:
        return off
    state_of = b.lti_state_of(g0)
    nent_of = np.diff(rowptr)

    def state_row(r):
        """(state, k, coefficient) when workspace row r is c * (row k of a state of the group)."""
        if nent_of[r] != 1:
            return None
        e = rowptr[r]
        if int(entbase[e]) not in state_of:
            return None
        return state_of[int(entbase[e])], int(entk[e]), float(entcoef[e])

    gt, gc = [], []
    for g in gterms:
        aoff, boff, nrows, wparam, doff, aimparam, flags = g[:7]
        if flags & GT_FLAG_DIAG:
            continue
        if flags != GT_FLAG_P or aoff != boff or nrows != N:
            return off
        rows = [state_row(r) for r in range(aoff, aoff + N)]
        if any(x is None for x in rows):
            return off
        st, c = rows[0][0], rows[0][2]
        if any(x != (st, k, c) for k, x in enumerate(rows)):
            return off
        gt.append([st * m * N, wparam, aimparam, doff])
        gc.append(c)
    if not 1 <= len(gt) <= T_SCAN_KMAX:
        return off
    # the column blocks: input j of the group as an unknown = N consecutive columns
    colblk = -np.ones(no, dtype=np.int64)
    blk = []
    for j, sid in enumerate(g0["ids"][:m]):
        spans = {(seg[4], seg[5]) for seg in b.segments if seg[6] == SEG_GATHER and seg[0] == sid}
        if len(spans) != 1:
            return off
        dst0, length = spans.pop()
        if length != N:
            return off
        if dst0 < ng:
            continue                                     # a given input: it enters d only
        if np.any(colblk[dst0 - ng:dst0 - ng + N] >= 0):
            return off
        colblk[dst0 - ng:dst0 - ng + N] = len(blk)
        blk.append([dst0 - ng, j * N])
    if not 1 <= len(blk) <= T_SCAN_BLKMAX:
        return off
    grow, gcoef, grest = -np.ones((nc, 2), dtype=np.int64), np.zeros(nc), []
    for R, axes in enumerate(g_rows):
        x = state_row(axes[0][0]) if len(axes) == 1 else None
        if x is None:
            grest.append(R)
            continue
        grow[R] = (x[0] * m * N + x[1], axes[0][1])
        gcoef[R] = x[2]
    # Fused set-up (H_T_SCAN_FUSED): the kernel can make its own table and its own d = Mg given -- no pre-pass, no
    # scratch -- when `given` is exactly the group's initial state in order (every state reads S at
    # S[k][j][i], j = the given column), every input of the group is an unknown, every workspace row is a row
    # of one of the K terms (d[first row + k] = c (A^{k+1} x0)[state]) and no row of G needs the column tables.
    rtot = len(rowptr) - 1
    sS = g0["ids"][m]
    s_spans = {(seg[1] % n, seg[2], seg[3], seg[4], seg[5]) for seg in b.segments if seg[6] == SEG_GATHER and seg[0] == sS}
    s_ok = bool(s_spans) and all(sp[1:] == (n * n, n, 0, n) for sp in s_spans) and all(
        seg[1] < n for seg in b.segments if seg[6] == SEG_GATHER and seg[0] == sS)
    fused = int(ng == n and s_ok and len(blk) == m and not grest and rtot == len(gt) * N
                and sorted(x[3] for x in gt) == [k * N for k in range(len(gt))])
    return dict(ok=1, K=len(gt), blk=np.asarray(blk, dtype=np.int64),
                gt=np.asarray(gt, dtype=np.int64), gc=np.asarray(gc, dtype=np.float64), grow=grow,
                gcoef=gcoef, grest=np.asarray(grest, dtype=np.int64), colblk=colblk,
                nother=int((colblk < 0).sum()), fused=fused)


def _sweep_tables(b, gterms, limit_recs, lax_recs, rowptr, entbase, entk, entcoef, group, nparams):
    """Tables of the sweep kernel (csrc/sweep.hip, plan_tables.h SW_*) for a dynamics compiled as
    ``ltv``: ``x+ = A_k x + B_k u`` with its own ``(A_k, B_k)`` per step and instance.  The kernel never
    forms a horizon matrix; with ``Phi(k, l) = A_k ... A_l`` (identity for ``l > k``) row k of an
    output ``c . x`` is ``c^T Phi(k, l+1) B_l`` in the columns of step l <= k (tools.py:27-31 with
    per-step matrices), and

        P[(j,l)][(j',l')] = B_l[:,j]^T Psi_l U[l][l'][:,j']            (l' <= l; the mirror image above)
        Psi_l = W_l + A_{l+1}^T Psi_{l+1} A_{l+1},   W_l = sum of w c c^T over the cost rows of step l
        q[(j,l)] = B_l[:,j]^T lam_l,   lam_l = rho_l + A_{l+1}^T lam_{l+1},   rho_l = sum w (d - aim) c

    -- O(n) multiply-adds per element of P and G, every row written once.  What the plan has to be
    for that: one such system (any number of axes sharing it), every unknown an input of it, every
    given value an initial state of it, every row of a cost or limit a fixed combination ``c`` of
    the states of ONE step and one axis, the steps of a row-set an arithmetic progression, no
    crossed cost.  Returns ``(tables, None)`` or ``(None, why not)``."""
    ng, no = b.ng, b.no
    n, m, N = group["n"], group["m"], group["N"]
    if n > SW_NMAX or m > SW_MMAX:
        return None, "more than %d states or %d inputs" % (SW_NMAX, SW_MMAX)
    state_of = b.lti_state_of(group)
    u_ids, s_id = group["ids"][:m], group["ids"][m]
    # axes: the bases of the group by the given columns of their initial state
    axis_of_x0, axes, base_axis = {}, [], {}
    for var, bid in b.base_ids.items():
        if bid not in state_of:
            continue
        segs = [b.segments[sg] for sg in sorted(set(int(x) for x in b.colseg[bid] if x >= 0))]
        x0 = [sg for sg in segs if sg[0] == s_id]
        ins = {sg[0]: sg for sg in segs if sg[0] in u_ids}
        if len(x0) != 1 or len(ins) != m or len(segs) != m + 1:
            return None, "a state of the system reads something else than its S and U_j"
        x0c, blocks = x0[0][4], [ins[sid][4] for sid in u_ids]
        if x0c >= ng or x0[0][5] != n or any(c < ng or ins[sid][5] != N for c, sid in zip(blocks, u_ids)):
            return None, "the initial state must be given and every input an unknown of N steps"
        rec = [x0c] + [c - ng for c in blocks]
        if x0c not in axis_of_x0:
            axis_of_x0[x0c] = len(axes)
            axes.append(rec)
        elif axes[axis_of_x0[x0c]] != rec:
            return None, "states of one axis with different columns"
        base_axis[bid] = axis_of_x0[x0c]
    if not 1 <= len(axes) <= SW_AXMAX:
        return None, "no axis, or more than %d" % SW_AXMAX
    col = -np.ones(no, dtype=np.int64)
    for a, rec in enumerate(axes):
        for j, c0 in enumerate(rec[1:]):
            if np.any(col[c0:c0 + N] >= 0):
                return None, "overlapping input columns"
            col[c0:c0 + N] = a | (j << 8) | (np.arange(N) << 16)
    if np.any(col < 0) or ng != n * len(axes) or sorted(r[0] for r in axes) != list(range(0, ng, n)):
        return None, "unknowns that are no input of the system, or given values that are no initial state"
    cvecs = []

    def rowset(row0, count, rs):
        """(axis, k0, kstep, cvec offset) of workspace rows row0 .. (one row for every line when rs == 1)."""
        rows = [row0] if rs == 1 else list(range(row0, row0 + count))
        info = []
        for r in rows:
            e0, e1 = rowptr[r], rowptr[r + 1]
            if e1 == e0:
                return None
            bids, ks = entbase[e0:e1], entk[e0:e1]
            if any(int(x) not in state_of for x in bids) or len(set(int(k) for k in ks)) != 1:
                return None
            ax = {base_axis[int(x)] for x in bids}
            if len(ax) != 1:
                return None
            c = np.zeros(n)
            for x, cf in zip(bids, entcoef[e0:e1]):
                c[state_of[int(x)]] += cf
            info.append((ax.pop(), int(ks[0]), c))
        a, k0, c = info[0]
        kstep = info[1][1] - k0 if len(info) > 1 else 0
        for i, (ai, ki, ci) in enumerate(info):
            if ai != a or ki != k0 + i * kstep or not np.array_equal(ci, c):
                return None
        if not (0 <= k0 < N and 0 <= k0 + (len(info) - 1) * kstep < N):
            return None
        for at, have in enumerate(cvecs):            # (the same combination: one entry)
            if np.array_equal(have, c):
                break
        else:
            at = len(cvecs)
            cvecs.append(c)
        return a, k0, (0 if rs == 1 else kstep), at * SW_NMAX

    terms = []
    for g in gterms:
        aoff, boff, nrows, wparam, doff, aimparam, flags = g[:7]
        if flags & GT_FLAG_DIAG:
            continue
        if flags != GT_FLAG_P or aoff != boff or doff != aoff:
            return None, "a crossed cost"
        info = rowset(aoff, nrows, nrows)
        if info is None:
            return None, "a cost whose rows are no fixed combination of one step's states"
        a, k0, ks, co = info
        if ks < 0:                 # (the rows of a cost in any order: steps ascending for the kernel)
            k0, ks = k0 + (nrows - 1) * ks, -ks
        terms.append([a, k0, ks, nrows, wparam, aimparam, co, 0])
    lims = []
    for out0, nrows, naxes, lax0, p_a, a_rows, p_c, c_rows, p_e, e_rows, *_ in limit_recs:
        if naxes > SW_AXMAX:
            return None, "a limit over more than %d axes" % SW_AXMAX
        rec = [out0, nrows, naxes, p_e, int(e_rows != 1), 0, 0, 0]
        for ax in range(SW_AXMAX):
            if ax >= naxes:
                rec += [0] * SW_LAX_WORDS
                continue
            off, rs = lax_recs[lax0 + ax]
            info = rowset(off, nrows, rs)
            if info is None:
                return None, "a limit whose rows are no fixed combination of one step's states"
            rec += [info[0], info[1], info[2], info[3], p_a + ax, (0 if a_rows == 1 else naxes),
                    p_c + ax, (0 if c_rows == 1 else naxes)]
        if any(rec[SW_LIM_WORDS + ax * SW_LAX_WORDS + 1:][:2] != rec[SW_LIM_WORDS + 1:][:2] for ax in range(naxes)):
            return None, "a limit whose axes take their lines from different steps"
        lims.append(rec)
    cv = np.zeros((len(cvecs), SW_NMAX))
    for i, c in enumerate(cvecs):
        cv[i, :n] = c
    axis_tab = np.zeros((len(axes), SW_AXIS_WORDS), dtype=np.int64)
    for a, rec in enumerate(axes):
        axis_tab[a, 0] = rec[0]
        axis_tab[a, 1:1 + m] = rec[1:]
    # what happens at every step, in lists the kernel walks (no search, no division per step):
    # the cost rows of step l, CENT[CPTR[l] .. CPTR[l+1]) = term; the lines of G of step l,
    # GENT[GPTR[l] .. GPTR[l+1]) = (limit, line i)
    cent = [[] for _ in range(N)]
    for ti, (a, k0, ks, cnt, *_rest) in enumerate(terms):
        for i in range(cnt):
            cent[k0 + i * ks].append(ti)
    gent = [[] for _ in range(N)]
    for li, rec in enumerate(lims):
        k0, ks = rec[SW_LIM_WORDS + 1], rec[SW_LIM_WORDS + 2]
        for i in range(rec[1]):
            gent[k0 + i * ks].append((li, i))
    cptr = np.cumsum([0] + [len(x) for x in cent])
    gptr = np.cumsum([0] + [len(x) for x in gent])
    cent = np.asarray([t for x in cent for t in x], dtype=np.int64)
    gent = np.asarray([t for x in gent for t in x], dtype=np.int64).reshape(-1, 2)
    return dict(n=n, m=m, N=N, src_a=group["ids"][0], src_b=group["ids"][1], axes=axis_tab,
                cptr=cptr, cent=cent, gptr=gptr, gent=gent,
                terms=np.asarray(terms, dtype=np.int64).reshape(-1, SW_TERM_WORDS),
                lims=np.asarray(lims, dtype=np.int64).reshape(-1, SW_LIM_WORDS + SW_AXMAX * SW_LAX_WORDS),
                col=col, cvec=cv.reshape(-1)), None


def _structural_patterns(form, b, fused, gterms, limit_recs, lax_recs, rtot, ldv, no, nc, groups=()):
    """Which entries of P (no x no) and of the stacked G (nc x no) can be non-zero at all:
    from the structurally non-zero elements of the workspace (an element with at least one
    op whose source is not a known zero -- ``U_j[k][l]`` above the diagonal l > k is zero
    for every system, tools.py:27-31; a source compiled as causal must stay so: ``Plan.causal_assumed``).
    None when the op lists were not built (huge plans)."""
    if rtot and fused["fd_idx"].size == 0:
        return None, None
    ops = fused["ops"].view(np.uint32).reshape(-1, 2)
    arena = fused["arena"].reshape(-1, 2)
    zero_src = np.zeros(int(fused["arena_total"]) + 1, dtype=bool)
    causal = causal_sources(form, b.sources, groups)
    for sid, src in enumerate(b.sources):
        if sid in causal:                                        # a U_j: (N, N, n), zeros above the diagonal
            N = src.array.shape[0]
            kk, ll = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
            above = np.repeat((ll > kk).reshape(-1), src.array.shape[2])
            off = int(arena[sid, 0])
            zero_src[off:off + above.size] = above
    live = ~zero_src[ops[:, 0].astype(np.int64)] if ops.size else np.zeros(0, dtype=bool)
    Vnz = np.zeros(max(rtot, 1) * ldv, dtype=bool)
    fd_idx, fd_ptr = fused["fd_idx"], fused["fd_ptr"]
    for i in range(fd_idx.size):
        Vnz[fd_idx[i]] = bool(live[fd_ptr[i]:fd_ptr[i + 1]].any())
    Vo = Vnz.reshape(-1, ldv)[:, :no]
    Pnz = np.zeros((no, no), dtype=bool)
    for g in gterms:
        if g[6] & GT_FLAG_DIAG:
            idx = np.arange(g[0], g[0] + g[2])
            Pnz[idx, idx] = True
        elif g[6] & GT_FLAG_P:
            a = Vo[g[0]:g[0] + g[2]].astype(np.int64)
            bb = Vo[g[1]:g[1] + g[2]].astype(np.int64)
            Pnz |= (a.T @ bb) > 0
    Gnz = np.zeros((nc, no), dtype=bool)
    for out0, nrows, naxes, lax0, *_ in limit_recs:
        for r in range(nrows):
            for ax in range(naxes):
                off, rs = lax_recs[lax0 + ax]
                Gnz[out0 + r] |= Vo[off + (0 if rs == 1 else r)]
    return Pnz, Gnz


def csc_pattern(mask, upper=False):
    """``(indptr, indices, flat)`` of a boolean pattern in CSC order (columns, rows ascending
    inside a column); ``flat[k] = row * ncols + col`` of the k-th stored entry.  ``upper``:
    only ``row <= col`` (what OSQP wants of P)."""
    mask = np.asarray(mask, dtype=bool)
    if upper:
        mask = np.triu(mask)
    cols_sorted, rows_sorted = np.nonzero(mask.T)        # column by column, rows ascending
    indptr = np.zeros(mask.shape[1] + 1, dtype=np.int32)
    np.cumsum(np.bincount(cols_sorted, minlength=mask.shape[1]), out=indptr[1:])
    flat = (rows_sorted.astype(np.int64) * mask.shape[1] + cols_sorted).astype(np.int32)
    return indptr, rows_sorted.astype(np.int32), flat


def compile_plan(form, costs=None, limits=None, lti=(), csc=None, workspace="auto", ltv=()):
    """Compile ``form`` (an up-to-date Formulation: sizes and IDs current).

    ``costs``: dict name -> Cost to include (default ``form.goals``);
    ``limits``: list of Constraint in stacking order (default: every limit of
    ``form.constraints`` then of ``form.constraint_boxes``, body.py:306-315);
    ``lti``: names of ExtendedSystem dynamics whose horizon matrices the assembly kernel
    generates on chip from per-instance ``(A, B)`` instead of reading ``S, U`` (K1 fused into
    the assembly; only the persistent kernel can run such a plan);
    ``ltv``: ONE name of an ExtendedSystem whose dynamics differ from step to step and instance to
    instance, ``x+ = A_k x + B_k u`` (BASELINE config C5): the sweep kernel (csrc/sweep.hip) takes
    ``A (N, n, n)``, ``B (N, n, m)`` per instance and assembles without forming a horizon matrix
    (:func:`_sweep_tables` states what the formulation has to be for that; ValueError otherwise).
    The reference has no such path (``dynamics.py:222-231`` re-extends ONE pair per tick): pinned to
    it where all steps share one pair;
    ``csc``: ``"upper"`` or ``"full"`` -- the plan's assembly writes, instead of dense P and G,
    the ``data`` arrays of their CSC forms on the structural pattern (P: its upper triangle /
    all of it), what ``scipy.sparse.csc_matrix(Q)``, ``csc_matrix(A)`` hand the solver in
    biped_mpc_loop.py:57-58: ``P`` becomes ``(B, plan.csc["pnnz"])``, ``G``
    ``(B, plan.csc["gnnz"])``; ``plan.csc["P"]``, ``plan.csc["G"]`` hold ``(indptr, indices)``.
    Only the persistent kernel writes this form: ValueError when the plan cannot run there.
    """
    b = _Builder(form)
    b.flatten_definitions()
    no = b.no

    if costs is None:
        costs = form.goals
    if limits is None:
        limits = [l for group in form.constraints.values() for l in group]
        limits += [l for box in form.constraint_boxes.values() for l in box.constraints]

    # ---- costs -> terms (body.py:266-302) ----------------------------------------
    terms = []
    for name, cost in costs.items():
        p_w = b.param(("cost", name, "weight"), lambda c=cost: [[float(c.weight)]])
        aim = np.asarray(cost.aim, dtype=np.float64)
        caim = np.asarray(cost.cross_aim, dtype=np.float64)
        if aim.ndim != 2 or aim.shape[0] != 1 or np.atleast_2d(caim).shape[0] != 1:
            # the reference fails to broadcast (no,1) += (no,r) for r > 1 (body.py:293-300)
            raise ValueError(
                "cost '{}': aim and cross_aim must hold one value per axis".format(name))
        p_aim = b.param(("cost", name, "aim"),
                        lambda c=cost: np.asarray(c.aim, dtype=np.float64).reshape(1, -1))
        crossed = bool(getattr(cost, "crossed", cost.cross != cost.variable))
        p_caim = (b.param(("cost", name, "cross_aim"),
                          lambda c=cost: np.asarray(c.cross_aim, dtype=np.float64).reshape(1, -1))
                  if crossed else p_aim)
        for i, axis in enumerate(cost.axes):
            va, nv = b.rowset(cost.variable + axis, cost.schedule, cost.L[i] if cost.L else None)
            ca, ncr = b.rowset(cost.cross + axis, cost.schedule,
                               cost.cross_L[i] if cost.cross_L else None)
            if nv != ncr:
                raise ValueError(
                    "cost '{}': {} rows of '{}' against {} rows of '{}'".format(
                        name, nv, cost.variable + axis, ncr, cost.cross + axis))
            diag = b.diagonal_rowset(va) if (not crossed and va == ca) else None
            terms.append(dict(va=va, ca=ca, n=nv, w=p_w, aim=p_aim + i, caim=p_caim + i,
                              plain=(not crossed and va == ca), diag=diag))

    # ---- limits (body.py:236-264, restrictions.py:147-199) -----------------------
    limit_recs, lax_recs, rowlimit, limit_rows = [], [], [], []
    out0 = 0
    for idx, limit in enumerate(limits):
        naxes = len(limit.axes)
        first = limit.variable + limit.axes[0]
        var_rows = b.var_matrix[first].shape[0]
        nlines = limit.nlines
        nrows = int(var_rows if nlines is None else nlines)
        lax0 = len(lax_recs)
        for i, axis in enumerate(limit.axes):
            off, rs = b.rowset(limit.variable + axis, limit.schedule,
                               limit.L[i] if limit.L else None)
            if rs not in (1, nrows):
                raise ValueError(
                    "constraint on '{}': {} rows cannot fill {} lines".format(
                        limit.variable + axis, rs, nrows))
            lax_recs.append([off, rs])          # off is a row-set id until place_rowsets
        arrow = np.asarray(limit.arrow, dtype=np.float64).reshape(-1, naxes)
        center = np.asarray(limit.center, dtype=np.float64).reshape(-1, naxes)
        extreme = np.asarray(limit.extreme, dtype=np.float64).reshape(-1, 1)
        for label, field in (("arrow", arrow), ("center", center), ("extreme", extreme)):
            if field.shape[0] not in (1, nrows):
                raise ValueError(
                    "constraint on '{}': '{}' has {} rows for {} lines".format(
                        limit.variable, label, field.shape[0], nrows))
        p_a = b.param(("limit", idx, "arrow"), lambda l=limit, k=naxes: np.asarray(
            l.arrow, dtype=np.float64).reshape(-1, k))
        p_c = b.param(("limit", idx, "center"), lambda l=limit, k=naxes: np.asarray(
            l.center, dtype=np.float64).reshape(-1, k))
        p_e = b.param(("limit", idx, "extreme"), lambda l=limit: np.asarray(
            l.extreme, dtype=np.float64).reshape(-1, 1))
        limit_recs.append([out0, nrows, naxes, lax0, p_a, arrow.shape[0], p_c, center.shape[0],
                           p_e, extreme.shape[0], 0, 0])
        rowlimit.extend([idx] * nrows)
        limit_rows.append((out0, nrows))
        out0 += nrows
    nc = out0

    # ---- place the row-sets some kernel reads; build the gterms -----------------------
    needed = set(rec[0] for rec in lax_recs)
    for t in terms:
        if t["diag"] is None:
            needed.update((t["va"], t["ca"]))
    where = b.place_rowsets(needed)
    for rec in lax_recs:
        rec[0] = where[rec[0]]
    gterms, diag_coefs = [], []
    for t in terms:
        if t["diag"] is not None:
            # P[c][c] += (w coef) coef,  q[c] += w (coef (0 - aim))  on c = col0 .. col0 + n - 1
            col0, coefs = t["diag"]
            gterms.append([col0, sum(len(c) for c in diag_coefs), t["n"], t["w"], 0, t["aim"],
                           GT_FLAG_DIAG, 0, 0, 0])
            diag_coefs.append(coefs)
        elif t["plain"]:
            #  q += w V^T (Vg g - aim)
            va = where[t["va"]]
            gterms.append([va, va, t["n"], t["w"], va, t["aim"], GT_FLAG_P, 0, 0, 0])
        else:
            #  P += w V^T C ;  q += w/2 V^T (Cg g - cross_aim) + w/2 C^T (Vg g - aim)
            va, ca = where[t["va"]], where[t["ca"]]
            gterms.append([va, ca, t["n"], t["w"], ca, t["caim"], GT_FLAG_P | GT_FLAG_HALF,
                           0, 0, 0])
            gterms.append([ca, -1, t["n"], t["w"], va, t["aim"], GT_FLAG_HALF, 0, 0, 0])
    diag_coefs = np.concatenate(diag_coefs) if diag_coefs else np.zeros(0)

    # ---- tables ---------------------------------------------------------------------
    rowptr, entbase, entk, entcoef, rtot = _csr_tables(
        b.placed_blocks, b.base_row0, b.base_rows, b.total_base_rows)
    assert rtot == b.rtot
    ldv = no + 2              # columns: the unknowns, d = Mg . given, a column of ones (persistent kernel)
    ldv += (2 - ldv) % 4      # = 2 mod 4
    fused = _fused_program(b, rowptr, entbase, entk, entcoef, rtot, ldv)
    # structural tile masks of the gterm operands (exact zeros of the workspace)
    tiles = fused["row_tiles"]
    for rec in gterms:
        if rec[6] & GT_FLAG_DIAG:
            continue

        def mask(off, n):
            m = 0
            for r in range(off, off + n):
                m |= int(tiles[r])
            return m
        rec[7] = mask(rec[0], rec[2])
        rec[8] = mask(rec[1], rec[2]) if rec[1] >= 0 else 0
    groups = _lti_groups(form, b.sources, lti)
    sweep = None
    if ltv:
        if len(tuple(ltv)) != 1 or lti or csc is not None:
            raise ValueError("ltv: one dynamics name, without lti= or csc=")
        sweep, why = _sweep_tables(b, gterms, limit_recs, lax_recs, rowptr, entbase, entk, entcoef,
                                   _lti_groups(form, b.sources, ltv)[0], len(b.params))
        if sweep is None:
            raise ValueError("dynamics %r cannot be assembled step by step: %s" % (tuple(ltv)[0], why))
    image = _resident_image(b.sources, b.ng, len(b.params), groups)
    # The persistent kernel's workspace: dense, or -- a wide problem whose row-sets each live in a
    # part of the columns -- compact, when that is what lets a second workgroup share the CU's LDS.
    # (Compact needs the 16-byte-piece paths of G: an even width, rows of at most two axes, no CSC.)
    ws = Workspace.dense(rtot, no, ldv)
    packed_like = csc is None and no % 2 == 0 and nc > 0 and all(rec[2] <= 2 for rec in limit_recs)
    if workspace not in ("auto", "dense", "compact"):
        raise ValueError("workspace: 'auto', 'dense' or 'compact'")
    if _os.environ.get("MPCASM_NO_COMPACT"):              # (A/B aids: tools/ab_workspace.py)
        workspace = "dense"
    elif _os.environ.get("MPCASM_COMPACT") and workspace == "auto":
        workspace = "compact"
    if packed_like and rtot > 0 and workspace != "dense" and (workspace == "compact"
                                                 or 8 * ws.doubles > RS_COMPACT_FROM_BYTES):
        cand = Workspace.windows(rtot, no, b.spans, fused["fd_idx"] // ldv, fused["fd_idx"] % ldv)
        if ((workspace == "compact" or 3 * cand.doubles <= 2 * ws.doubles)
                and (cand.row0 + rtot + 8) * cand.ldv < 65536 and len(b.params) < 65535):
            ws = cand
    resident = _resident_program(fused, gterms, no, ldv, ws, image, b.ng, len(b.params), nc)
    rs_rr = _resident_rows(limit_recs, lax_recs, len(b.params), ws)
    # per row of G the windows of its (at most two) axes, in column pairs: first | count << 8 of the
    # first axis, the same << 16 of the second (compact workspace only; a missing axis: 0 | 0)
    rs_rrwin = np.zeros(0, dtype=np.int32)
    if ws.compact:
        win = []
        for out0, nrows, naxes, lax0, *_ in limit_recs:
            for r in range(nrows):
                word = 0
                for ax in range(naxes):
                    off, rs = lax_recs[lax0 + ax]
                    row = off + (0 if rs == 1 else r)
                    word |= (int(ws.c0[row]) // 2 | (int(ws.w[row]) // 2) << 8) << (16 * ax)
                win.append(word)
        rs_rrwin = np.asarray(win, dtype=np.int64).astype(np.uint32).view(np.int32)
    if any(rec[2] > RS_AXMAX for rec in limit_recs):
        rs_rr = np.zeros(0, dtype=np.int32)      # too many axes: no resident kernel
    pm_blocks, pm_rows, r0 = [], {}, 0
    for var in form.definitions.keys():
        M = b.var_matrix[var]
        pm_blocks.append(M)
        pm_rows[var] = (r0, M.shape[0])
        r0 += M.shape[0]
    pm_rowptr, pm_entbase, pm_entk, pm_entcoef, pmrows = _csr_tables(
        pm_blocks, b.base_row0, b.base_rows, b.total_base_rows)

    sections = [
        ("OFF_SEG", np.asarray(b.segments, dtype=np.int32).reshape(-1)),
        ("OFF_COLSEG", (np.concatenate(b.colseg) if b.colseg else np.zeros(0)).astype(np.int32)),
        ("OFF_ROWPTR", rowptr),
        ("OFF_ENTBASE", entbase),
        ("OFF_ENTK", entk),
        ("OFF_GTERM", np.asarray(gterms, dtype=np.int32).reshape(-1)),
        ("OFF_LIMIT", np.asarray(limit_recs, dtype=np.int32).reshape(-1)),
        ("OFF_LAX", np.asarray(lax_recs, dtype=np.int32).reshape(-1)),
        ("OFF_ROWLIMIT", np.asarray(rowlimit, dtype=np.int32)),
        ("OFF_PM_ROWPTR", pm_rowptr),
        ("OFF_PM_ENTBASE", pm_entbase),
        ("OFF_PM_ENTK", pm_entk),
        ("OFF_ARENA", fused["arena"]),
        ("OFF_FD_IDX", fused["fd_idx"]),
        ("OFF_FD_PTR", fused["fd_ptr"]),
        ("OFF_OP", fused["ops"]),
        ("OFF_RS_PROG", _packed_program(resident)),
        ("OFF_RS_SRC", resident["src"]),
        ("OFF_RS_GIDX", resident["gidx"]),
        ("OFF_RS_DST", resident["dst"]),
        ("OFF_RS_TRIP", resident["trips"]),
        ("OFF_RS_WTRIP", resident["wtrip"]),
        ("OFF_RS_SPLIT", resident["split"]),
        ("OFF_RS_ZBLK", resident["zblk"]),
        ("OFF_RS_RR", rs_rr),
        ("OFF_RS_INMETA", image["meta"]),
        ("OFF_RS_ABMETA", image["ab_meta"]),
        ("OFF_RS_LTI", np.asarray(
            [[g["n"], g["m"], g["N"], g["img_a"], g["img_b"], g["tab_a"], g["tab_b"], g["tab_p"]]
             for g in groups], dtype=np.int32).reshape(-1)),
    ]
    # what the persistent kernel would otherwise derive once per workgroup, ready to copy:
    # per column of the unknowns the (weight, aim) slots and coefficients of the (at most
    # RS_DIAG_MAX = 2) diagonal gterms on it (free slots: the always-zero parameter), ...
    rs_dpar = np.full((no, 2 * RS_DIAG_MAX), len(b.params), dtype=np.int32)
    rs_dcoef = np.zeros((no, RS_DIAG_MAX))
    taken = np.zeros(no, dtype=np.int64)
    for g in gterms:
        if g[6] & GT_FLAG_DIAG:
            for k in range(g[2]):
                c, j = g[0] + k, int(taken[g[0] + k])
                if j < RS_DIAG_MAX:
                    rs_dpar[c, 2 * j:2 * j + 2] = [g[3], g[5]]
                    rs_dcoef[c, j] = diag_coefs[g[1] + k]
                taken[c] += 1
    # ... and, for small problems, the descriptor of every 16-byte piece of G a stream-wave
    # thread owns: piece e = t + u RS_GDESC_THREADS, columns 2cp, 2cp+1 of row R = e // (no/2):
    # Workspace.index(row0, 2cp) | Workspace.index(row1, 2cp) << 16, arrow0 | arrow1 << 16
    rs_gdesc = rs_gfix = np.zeros(0, dtype=np.int32)
    rs_ngfix = 0
    rs_gsingle = 0
    rr_ok = rs_rr.size == nc * RS_RR_WORDS
    packed_ok = (rr_ok and no % 2 == 0 and nc > 0 and (ws.row0 + b.rtot + 8) * ws.ldv < 65536
                 and len(b.params) < 65535
                 and bool((rs_rr.reshape(nc, RS_RR_WORDS)[:, 12] <= 2).all()))
    assert packed_ok or not ws.compact
    if packed_ok and nc * (no // 2) <= RS_GDESC_PIECES * RS_GDESC_THREADS:
        recs = rs_rr.reshape(nc, RS_RR_WORDS).astype(np.int64)
        e = np.arange(RS_GDESC_PIECES * RS_GDESC_THREADS)
        live = e < nc * (no // 2)
        R = np.where(live, e // (no // 2), 0)
        cp = np.where(live, e % (no // 2), 0)
        # Which of a piece's (at most two) axes can be non-zero in its two columns at all: an
        # element of the workspace nothing is composed into is an exact zero.  The axis that can
        # goes first; where all 64 pieces of one wavefront's round have at most one such axis
        # (the usual case: a variable per axis, columns partitioned by axis) the kernel reads one
        # workspace row and one arrow instead of two (bit u * 4 + wave of RS_GSINGLE).
        nz = np.zeros(max(rtot, 1) * ldv, dtype=bool)
        nz[fused["fd_idx"]] = True
        nz = nz.reshape(-1, ldv)

        def row_of(voff):                                # voff = Workspace.rowstart(row)
            return np.clip(voff // ws.ldv - ws.row0, 0, max(rtot, 1) - 1)

        def can(row, col):
            return nz[row, col] | nz[row, col + 1]

        def inside(row, col):                            # (windows are whole 4-column blocks)
            return (ws.c0[row] <= col) & (col + 2 <= ws.c0[row] + ws.w[row])

        v0, v1, a0, a1 = recs[R, 0], recs[R, 1], recs[R, 4], recs[R, 5]
        two = recs[R, 12] >= 2
        r0, r1 = row_of(v0), row_of(v1)
        c0, c1 = can(r0, 2 * cp), two & can(r1, 2 * cp)
        # the piece in the row's window: its index; outside (compact workspace): row 0, zero arrow
        in0, in1 = inside(r0, 2 * cp), two & inside(r1, 2 * cp)
        p0 = np.where(in0, v0 + 2 * cp - ws.c0[r0], 0)
        p1 = np.where(in1 | ~two, v1 + np.where(two, 2 * cp - ws.c0[r1], 0), 0)
        a0, a1 = np.where(in0, a0, len(b.params)), np.where(in1 | ~two, a1, len(b.params))
        if not ws.compact:
            p1 = v1 + 2 * cp                             # (a missing axis: row 0 + the piece's columns)
        swap = c1 & ~c0                                  # only the second axis can: it goes first
        p0, p1 = np.where(swap, p1, p0), np.where(swap, p0, p1)
        a0, a1 = np.where(swap, a1, a0), np.where(swap, a0, a1)
        single = ~(c0 & c1) | ~live
        word0 = p0 | (p1 << 16)                          # Workspace.index(row, 2 cp)
        word1 = a0 | (a1 << 16)
        rs_gdesc = np.stack([word0, word1], axis=1).astype(np.uint32).view(np.int32).reshape(-1)
        # A few pieces with two live axes among many with one (the biped's 34-wide phase: x ends
        # and y begins inside one piece of every two-axis row) would turn every round they sit in
        # into a two-axis round.  Instead the round stays a one-axis round and adds the second axis
        # for that lane alone; a thread holds the second axis of at most one of its pieces (RS_GFIX).
        # (Writing such pieces apart, behind the rounds, leaves 16-byte holes in the lines the
        # rounds write: partial lines cost the write stream half its rate, tools/run_variant.py.)
        both = np.flatnonzero(~single)
        in_mixed_rounds = int((~single.reshape(-1, 64).all(axis=1)).sum()) * 64
        owners = both % RS_GDESC_THREADS
        if (both.size and 4 * both.size <= in_mixed_rounds and np.unique(owners).size == both.size
                and not _os.environ.get('MPCASM_NO_GFIX')):            # (A/B aid: tools/run_variant.py)
            fix = np.zeros((RS_GDESC_THREADS, 2), dtype=np.int64)
            fix[:, 0], fix[:, 1] = RS_GFIX_NONE << 16, len(b.params)
            fix[owners, 0] = p1[both] | ((both // RS_GDESC_THREADS) << 16)
            fix[owners, 1] = a1[both]
            rs_gfix, rs_ngfix = fix.astype(np.uint32).view(np.int32).reshape(-1), int(both.size)
            single = np.ones_like(single)
        rounds = single.reshape(RS_GDESC_PIECES, RS_GDESC_THREADS // 64, 64).all(axis=2)
        rs_gsingle = int(sum(1 << (u * (RS_GDESC_THREADS // 64) + w)
                             for u in range(RS_GDESC_PIECES) for w in range(RS_GDESC_THREADS // 64)
                             if rounds[u, w]))
    # ---- CSC hand-off: where the stored entries sit (P: in the LDS copy of P, leading
    # dimension no rounded up to even) and, per stored entry (R, c) of G, what a 16-byte piece
    # has for two columns: Workspace.index(row0, c) | Workspace.index(row1, c) << 16, arrow0 | arrow1 << 16
    P_pattern, G_pattern = _structural_patterns(
        form, b, fused, gterms, limit_recs, lax_recs, rtot, ldv, no, nc, groups)
    csc_p = csc_g = np.zeros(0, dtype=np.int32)
    csc_info, csc_gsingle = None, 0
    if csc is not None:
        if csc not in ("upper", "full"):
            raise ValueError("csc: 'upper', 'full' or None")
        two_axes = (rr_ok and (b.rtot + 8) * ldv < 65536 and len(b.params) < 65535
                    and bool((rs_rr.reshape(nc, RS_RR_WORDS)[:, 12] <= 2).all()))
        if not (resident["ok"] and two_axes):
            raise ValueError("this plan does not run on the persistent kernel: assemble dense "
                             "and convert with export_csc")
        pm = np.ones((no, no), dtype=bool) if P_pattern is None else P_pattern
        gm = np.ones((nc, no), dtype=bool) if G_pattern is None else G_pattern
        p_indptr, p_indices, p_flat = csc_pattern(pm, csc == "upper")
        g_indptr, g_indices, g_flat = csc_pattern(gm)
        ldp = no + (no & 1)
        csc_p = ((p_flat // no) * ldp + p_flat % no).astype(np.int32)
        recs = rs_rr.reshape(nc, RS_RR_WORDS).astype(np.int64)
        R, c = g_flat.astype(np.int64) // no, g_flat.astype(np.int64) % no
        nz = np.zeros(max(rtot, 1) * ldv, dtype=bool)
        nz[fused["fd_idx"]] = True
        if fused["fd_idx"].size == 0:
            nz[:] = True
        v0, v1, a0, a1 = recs[R, 0], recs[R, 1], recs[R, 4], recs[R, 5]
        c0, c1 = nz[v0 + c], (recs[R, 12] >= 2) & nz[v1 + c]
        swap = c1 & ~c0                                  # only the second axis can be non-zero
        v0, v1 = np.where(swap, v1, v0), np.where(swap, v0, v1)
        a0, a1 = np.where(swap, a1, a0), np.where(swap, a0, a1)
        csc_g = np.stack([(v0 + c) | ((v1 + c) << 16), a0 | (a1 << 16)], axis=1) \
            .astype(np.uint32).view(np.int32).reshape(-1)
        csc_gsingle = int(bool((~(c0 & c1)).all()))
        csc_info = dict(kind=csc, pnnz=int(p_flat.size), gnnz=int(g_flat.size),
                        P=(p_indptr, p_indices), G=(g_indptr, g_indices),
                        p_flat=p_flat, g_flat=g_flat)
    pmprog = _preview_program(b, pm_rowptr, pm_entbase, pm_entk, pm_entcoef, pmrows)
    sections += [("OFF_CSC_P", csc_p), ("OFF_CSC_G", csc_g)]
    # ---- column tables + tiled program (wide problems) ------------------------------------
    g_rows = []           # per row of G: (workspace row, arrow's parameter slot) of every axis
    for out0, nrows, naxes, lax0, p_a, a_rows, *_ in limit_recs:
        for r in range(nrows):
            g_rows.append([(lax_recs[lax0 + ax][0] + (0 if lax_recs[lax0 + ax][1] == 1 else r),
                            p_a + (0 if a_rows == 1 else r) * naxes + ax) for ax in range(naxes)])
    tiled = _tiled_program(b, form, gterms, rowptr, entbase, entk, entcoef, rtot, groups, rr_ok,
                           g_rows)
    scan = _scan_tables(b, gterms, rowptr, entbase, entk, entcoef, groups, g_rows, tiled, len(b.params))
    tiled["scan"] = scan
    ndt0 = (entcoef.size + pm_entcoef.size + fused["coefpool"].size + resident["coef"].size
            + diag_coefs.size)
    doff_delta = ndt0 + (ndt0 & 1) + 4 + rs_dcoef.size + pmprog["pool"].size
    ci = tiled["ci"]
    own = (ci[:, :, 1] >> 24) == T_SID_CONST          # offsets into the plan's own dtab
    ci[:, :, 0] += np.where(own, doff_delta, 0)
    ci32 = ci.astype(np.uint32).view(np.int32) if ci.size else np.zeros((0, 0, 2), np.int32)
    sections += [("OFF_T_CIG", np.ascontiguousarray(ci32[:, :b.ng]).reshape(-1)),
                 ("OFF_T_CIO", np.ascontiguousarray(ci32[:, b.ng:]).reshape(-1)),
                 ("OFF_T_STAGE", tiled["stages"].astype(np.uint32).view(np.int32).reshape(-1)),
                 ("OFF_T_LTI", tiled["lti"].astype(np.int32).reshape(-1)),
                 ("OFF_T_LTI_IDS", tiled["lti_ids"].astype(np.int32))]
    t_grow = -np.ones((nc, RS_AXMAX), dtype=np.int32)
    if rr_ok:
        for out0, nrows, naxes, lax0, *_ in limit_recs:
            for r in range(nrows):
                for ax in range(naxes):
                    off, rs = lax_recs[lax0 + ax]
                    t_grow[out0 + r, ax] = off + (0 if rs == 1 else r)
    sections += [("OFF_T_GROW", t_grow.reshape(-1)),
                 ("OFF_T_SROW", tiled["srow"].astype(np.int32)),
                 ("OFF_T_PIG", tiled["pig"].astype(np.int32)),
                 ("OFF_T_GREST", tiled["grest"].astype(np.int32)),
                 ("OFF_T_BROW0", np.asarray(list(b.base_row0) + [b.total_base_rows], dtype=np.int32))]
    # the columns some segment of every base variable covers (everywhere else its rows are zero)
    bcols = [np.flatnonzero(colseg >= 0) for colseg in b.colseg]
    sections += [("OFF_T_BCOLPTR", np.cumsum([0] + [c.size for c in bcols]).astype(np.int32)),
                 ("OFF_T_BCOLS", (np.concatenate(bcols) if bcols else np.zeros(0)).astype(np.int32))]
    # f2 (preview.hip): the same column tables unrolled per base row -- entry (element offset in its
    # stream, column of [given | unknowns] | stream << 24) for every covered column -- and, per entry of
    # a definition's row, where its base row sits among all base rows: what a workgroup copies into LDS
    # once and then reads per instance with no table look-up left
    p1ptr, p1ent = np.zeros(1, dtype=np.int64), []
    if tiled["ci_ok"]:
        for bid, cols in enumerate(bcols):
            k = np.arange(b.base_rows[bid], dtype=np.int64)[:, None]
            word = ci[bid, cols, 1]
            step = ((word & 0xFFFFFF) ^ 0x800000) - 0x800000              # (24 bits, signed)
            offs = ci[bid, cols, 0][None, :] + k * step[None, :]
            tag = (cols | ((word >> 24) << 24))[None, :].repeat(k.shape[0], axis=0)
            p1ent.append(np.stack([offs, tag], axis=2).reshape(-1, 2))
        p1ptr = np.concatenate([[0], np.cumsum([len(bcols[bid]) for bid in range(len(bcols))
                                                for _ in range(b.base_rows[bid])])]).astype(np.int64)
    p1ent = np.concatenate(p1ent) if p1ent else np.zeros((0, 2), dtype=np.int64)
    p1_ok = bool(tiled["ci_ok"]) and (p1ent.size == 0 or (0 <= p1ent[:, 0].min() and p1ent[:, 0].max() < 1 << 31))
    if not p1_ok:
        p1ptr, p1ent = np.zeros(1, dtype=np.int64), np.zeros((0, 2), dtype=np.int64)
    p2y = (np.asarray(b.base_row0, dtype=np.int64)[pm_entbase] + pm_entk) if p1_ok and pm_entbase.size \
        else np.zeros(0, dtype=np.int64)
    sections += [("OFF_T_P1PTR", np.asarray(p1ptr, dtype=np.int32)),
                 ("OFF_T_P1ENT", p1ent.astype(np.uint32).view(np.int32).reshape(-1)),
                 ("OFF_T_P2Y", p2y.astype(np.int32))]
    sw_empty = dict(axes=np.zeros(0), terms=np.zeros(0), lims=np.zeros(0), col=np.zeros(0), cvec=np.zeros(0),
                    cptr=np.zeros(0), cent=np.zeros(0), gptr=np.zeros(0), gent=np.zeros(0))
    sw = sweep or sw_empty
    sections += [("OFF_SW_AXIS", np.asarray(sw["axes"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_TERM", np.asarray(sw["terms"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_LIM", np.asarray(sw["lims"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_COL", np.asarray(sw["col"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_CPTR", np.asarray(sw["cptr"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_CENT", np.asarray(sw["cent"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_GPTR", np.asarray(sw["gptr"]).astype(np.int32).reshape(-1)),
                 ("OFF_SW_GENT", np.asarray(sw["gent"]).astype(np.int32).reshape(-1))]
    sections += [("OFF_T_SCAN_BLK", scan["blk"].astype(np.int32).reshape(-1)),
                 ("OFF_T_SCAN_GT", scan["gt"].astype(np.int32).reshape(-1)),
                 ("OFF_T_SCAN_GROW", scan["grow"].astype(np.int32).reshape(-1)),
                 ("OFF_T_SCAN_GREST", scan["grest"].astype(np.int32)),
                 ("OFF_T_SCAN_COLBLK", scan["colblk"].astype(np.int32))]
    sections += [("OFF_RS_DPAR", rs_dpar.reshape(-1)), ("OFF_RS_GDESC", rs_gdesc),
                 ("OFF_RS_GFIX", rs_gfix), ("OFF_RS_RRWIN", rs_rrwin),
                 ("OFF_PM_MAP", pmprog["map"]), ("OFF_PM_FDPTR", pmprog["fd_ptr"]),
                 ("OFF_PM_OP", pmprog["ops"])]
    header = np.zeros(H_WORDS, dtype=np.int32)
    parts, off = [header], H_WORDS
    for name, arr in sections:
        if name == "OFF_OP" and off & 1:          # the kernels read ops as 8-byte pairs
            parts.append(np.zeros(1, dtype=np.int32))
            off += 1
        if name == "OFF_PM_OP" and off & 1:       # ... 8-byte pairs
            parts.append(np.zeros(1, dtype=np.int32))
            off += 1
        if name in ("OFF_RS_RR", "OFF_RS_INMETA", "OFF_RS_ABMETA", "OFF_RS_DPAR", "OFF_RS_PROG",
                    "OFF_RS_GDESC", "OFF_RS_GFIX", "OFF_CSC_G", "OFF_T_CIG", "OFF_T_CIO", "OFF_T_P1ENT",
                    "OFF_T_STAGE", "OFF_T_GROW", "OFF_T_SROW", "OFF_T_PIG", "OFF_T_SCAN_BLK",
                    "OFF_T_SCAN_GT", "OFF_T_SCAN_GROW", "OFF_SW_AXIS", "OFF_SW_TERM", "OFF_SW_LIM") and off & 3:   # ... 16-byte quads
            pad = 4 - (off & 3)
            parts.append(np.zeros(pad, dtype=np.int32))
            off += pad
        if name == "OFF_RS_TRIP" and off & 7:     # ... 32-byte records (one scalar load each)
            pad = 8 - (off & 7)
            parts.append(np.zeros(pad, dtype=np.int32))
            off += pad
        header[_H[name]] = off
        parts.append(arr)
        off += arr.size
    dparts = [entcoef, pm_entcoef, fused["coefpool"], resident["coef"], diag_coefs]
    ndt = sum(part.size for part in dparts)
    dparts.append(np.zeros(ndt & 1))             # the constant stream starts 16-byte aligned
    header[_H["DOFF_RS_CONST"]] = ndt + (ndt & 1)
    dparts.append(np.array([1.0, 1.0, 0.0, 0.0]))
    header[_H["DOFF_RS_DCOEF"]] = ndt + (ndt & 1) + 4
    dparts.append(rs_dcoef.reshape(-1))
    header[_H["RS_NGDESC"]] = rs_gdesc.size // 2
    header[_H["RS_NGFIX"]] = rs_ngfix
    header[_H["RS_COMPACT"]], header[_H["RS_LDV"]] = ws.compact, ws.ldv
    header[_H["RS_VD"]], header[_H["RS_VROW0"]] = ws.vd, ws.row0
    header[_H["DOFF_PM_POOL"]] = ndt + (ndt & 1) + 4 + rs_dcoef.size
    dparts.append(pmprog["pool"])
    header[_H["PM_NFD"]], header[_H["PM_NOPS"]] = pmprog["nfd"], pmprog["ops"].size // 2
    header[_H["PM_NPOOL"]] = pmprog["pool"].size
    assert sum(part.size for part in dparts) == doff_delta
    dparts.append(tiled["delta"])
    header[_H["T_DOFF_DELTA"]], header[_H["T_NDELTA"]] = doff_delta, tiled["delta"].size
    header[_H["T_DOFF_SCOEF"]] = doff_delta + tiled["delta"].size
    dparts.append(tiled["scoef"])
    header[_H["T_DOFF_SCAN_GC"]] = doff_delta + tiled["delta"].size + tiled["scoef"].size
    dparts.append(scan["gc"])
    header[_H["T_DOFF_SCAN_GCOEF"]] = header[_H["T_DOFF_SCAN_GC"]] + scan["gc"].size
    dparts.append(scan["gcoef"])
    header[_H["SW_DOFF_CVEC"]] = header[_H["T_DOFF_SCAN_GCOEF"]] + scan["gcoef"].size
    header[_H["SW_DOFF_CVEC"]] += header[_H["SW_DOFF_CVEC"]] & 1          # (read as 16-byte pairs)
    dparts.append(np.zeros(int(header[_H["SW_DOFF_CVEC"]]) - sum(part.size for part in dparts)))
    dparts.append(np.asarray(sw["cvec"], dtype=np.float64))
    header[_H["SW_NCVEC"]] = np.asarray(sw["cvec"]).size // SW_NMAX
    if sweep is not None:
        header[_H["SW_OK"]], header[_H["SW_N"]], header[_H["SW_M"]] = 1, sweep["n"], sweep["m"]
        header[_H["SW_HORIZON"]], header[_H["SW_NAXES"]] = sweep["N"], sweep["axes"].shape[0]
        header[_H["SW_SRC_A"]], header[_H["SW_SRC_B"]] = sweep["src_a"], sweep["src_b"]
        header[_H["SW_NTERM"]], header[_H["SW_NLIM"]] = sweep["terms"].shape[0], sweep["lims"].shape[0]
        header[_H["SW_NCENT"]], header[_H["SW_NGENT"]] = sweep["cent"].size, sweep["gent"].shape[0]
    header[_H["T_SCAN"]] = scan["K"] if scan["ok"] else 0
    header[_H["T_SCAN_NBLK"]], header[_H["T_SCAN_NGREST"]] = scan["blk"].shape[0], scan["grest"].size
    header[_H["T_SCAN_NOTHER"]] = scan["nother"]
    header[_H["T_SCAN_FUSED"]] = scan["fused"]
    header[_H["T_NGREST"]] = tiled["grest"].size
    header[_H["T_CI_OK"]], header[_H["T_NOP"]] = tiled["ci_ok"], tiled["nop"]
    header[_H["T_OK"]], header[_H["T_NSTAGE"]] = tiled["ok"], tiled["stages"].shape[0]
    header[_H["T_NLTI"]], header[_H["T_WORK"]] = tiled["lti"].shape[0], tiled["work"]
    header[_H["T_TOEPLITZ"]] = tiled["toeplitz"]
    header[_H["T_NP1"]] = p1ent.shape[0] if p1_ok else -1
    dtab = np.concatenate(dparts).astype(np.float64)
    params = np.asarray(b.params, dtype=np.float64)
    header[_H["MAGIC"]], header[_H["VERSION"]] = PLAN_MAGIC, PLAN_VERSION
    header[_H["NG"]], header[_H["NO"]], header[_H["NC"]] = b.ng, no, nc
    header[_H["NPARAMS"]] = params.size
    header[_H["NSRC"]], header[_H["NBASE"]] = len(b.sources), len(b.base_rows)
    header[_H["NSEG"]], header[_H["RTOT"]], header[_H["NENT"]] = len(b.segments), rtot, entcoef.size
    header[_H["NGTERM"]], header[_H["NLIMIT"]] = len(gterms), len(limit_recs)
    header[_H["NLAX"]], header[_H["PMROWS"]] = len(lax_recs), pmrows
    header[_H["PM_NENT"]], header[_H["LDV"]] = pm_entcoef.size, ldv
    header[_H["DOFF_ENTCOEF"]], header[_H["DOFF_PM_ENTCOEF"]] = 0, entcoef.size
    header[_H["FUSED_OK"]], header[_H["ARENA_TOTAL"]] = fused["ok"], fused["arena_total"]
    header[_H["NFD"]], header[_H["NOPS"]] = fused["fd_idx"].size, fused["ops"].size // 2
    header[_H["NCOEF"]] = fused["coefpool"].size
    header[_H["DOFF_COEFPOOL"]] = entcoef.size + pm_entcoef.size
    header[_H["RS_OK"]], header[_H["RS_JC"]] = resident["ok"], resident["jc"]
    header[_H["RS_SYM"]] = resident["sym"]
    header[_H["RS_NTRIP"]] = resident["ntrip"]
    header[_H["RS_GSINGLE"]] = rs_gsingle
    header[_H["CSC_PNNZ"]], header[_H["CSC_GNNZ"]] = csc_p.size, csc_g.size // 2
    header[_H["CSC_GSINGLE"]] = csc_gsingle
    header[_H["RS_NZBLK"]] = resident["zblk"].size
    header[_H["RS_NSPLIT"]] = resident["split"].size
    header[_H["RS_UNIT"]], header[_H["RS_NCHUNK"]] = image["unit"], image["nchunk"]
    header[_H["RS_IMG"]] = image["img"]
    header[_H["RS_IMG_DMA"]] = image["dma"]
    header[_H["RS_NLTI"]] = len(groups)
    header[_H["RS_AB"]] = image["ab"]
    header[_H["RS_IMG_GIVEN"]], header[_H["RS_IMG_PARAMS"]] = image["given"], image["params"]
    if rs_rr.size != nc * RS_RR_WORDS:
        header[_H["RS_OK"]] = 0                  # a constraint with more than RS_AXMAX axes
    else:                                        # G by 16-byte pieces from the packed words
        header[_H["RR_PACKED"]] = int(packed_ok)
    header[_H["DOFF_RS_COEF"]] = entcoef.size + pm_entcoef.size + fused["coefpool"].size
    header[_H["DOFF_DIAGCOEF"]] = (entcoef.size + pm_entcoef.size + fused["coefpool"].size
                                   + resident["coef"].size)
    header[_H["NDIAGCOEF"]] = diag_coefs.size
    header[_H["NITAB"]], header[_H["NDTAB"]] = off, dtab.size

    plan = Plan()
    plan.itab = np.concatenate(parts).astype(np.int32)
    plan.dtab = dtab
    plan.ng, plan.no, plan.nc = b.ng, no, nc
    plan.sources = b.sources
    plan.params = params
    plan.param_slots = b.param_slots
    plan.param_getters = b.param_getters
    plan.fingerprint = structure_fingerprint(costs, limits)
    plan.pm_rows, plan.pmrows = pm_rows, pmrows
    plan.rtot, plan.ldv, plan.workspace = rtot, ldv, ws
    plan.limit_rows = limit_rows
    plan.optim_ID = {v: form.optim_ID[v] for v in form.optim_variables}
    plan.given_ID = {v: form.given_ID[v] for v in form.given_variables}
    plan.n_gterms = len(gterms)
    plan.resident = resident
    plan.P_pattern, plan.G_pattern = P_pattern, G_pattern
    plan.csc = csc_info
    plan.lti = [dict(name=g["name"], n=g["n"], m=g["m"], N=g["N"], ids=list(g["ids"])) for g in groups]
    plan.ltv = ([dict(name=tuple(ltv)[0], n=sweep["n"], m=sweep["m"], N=sweep["N"],
                      ids=[sweep["src_a"], sweep["src_b"]])] if sweep is not None else [])
    plan.sweep = sweep
    plan.tiled = tiled
    # Sources (U_j read from memory) whose zeros above the diagonal some table of this plan relies
    # on -- the tile masks and stage classes of the tiled kernel, the CSC patterns: whatever is
    # bound in their place must be causal too (Assembler.bind_source / rebind_sources check it).
    generated = {i for g in groups for i in g["ids"]}
    plan.causal_assumed = sorted(i for i in tiled["causal"] if i not in generated) \
        if (tiled["ok"] or csc is not None) else []
    return plan
