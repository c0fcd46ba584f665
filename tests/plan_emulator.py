"""numpy interpreter of the plan tables  --  TEST INFRASTRUCTURE ONLY.

Executes ``itab`` / ``dtab`` exactly as csrc/plan_tables.h documents them, one
instance at a time, so that the host plan compiler (``mpcasm.plan``) can be
checked against the oracle on a machine without a GPU.  It is not a fallback:
nothing under ``mpc-interface_amd/`` imports it.
"""
import numpy as np

from mpcasm import plan as P

H = P._H


def _section(itab, name, count):
    off = itab[H[name]]
    return itab[off:off + count]


def _row(plan, srcs, rowptr, entbase, entk, coef, r, W):
    it = plan.itab
    W_ = W
    segs = _section(it, "OFF_SEG", it[H["NSEG"]] * P.SEG_WORDS).reshape(-1, P.SEG_WORDS)
    colseg = _section(it, "OFF_COLSEG", it[H["NBASE"]] * W_).reshape(-1, W_) if W_ else None
    out = np.zeros(W_)
    for e in range(rowptr[r], rowptr[r + 1]):
        u, k, cf = entbase[e], entk[e], coef[e]
        for c in range(W_):
            sg = colseg[u, c]
            if sg < 0:
                continue
            s = segs[sg]
            j = c - s[4]
            if s[6] == P.SEG_IDENTITY:
                val = 1.0 if j == k else 0.0
            else:
                val = srcs[s[0]].ravel()[s[1] + k * s[2] + j * s[3]]
            out[c] += cf * val
    return out


def run(plan, given, params=None, sources=None):
    """Returns dict with V, P, q, G, h, PM for one instance."""
    it, dt = plan.itab, plan.dtab
    ng, no, nc = plan.ng, plan.no, plan.nc
    W = ng + no
    srcs = [s.array for s in plan.sources] if sources is None else sources
    params = plan.params if params is None else params
    g = np.asarray(given, dtype=float).ravel()

    rtot, nent = it[H["RTOT"]], it[H["NENT"]]
    rowptr = _section(it, "OFF_ROWPTR", rtot + 1)
    entbase, entk = _section(it, "OFF_ENTBASE", nent), _section(it, "OFF_ENTK", nent)
    coef = dt[it[H["DOFF_ENTCOEF"]]:it[H["DOFF_ENTCOEF"]] + nent]
    V = np.zeros((rtot, no + 1))
    for r in range(rtot):
        row = _row(plan, srcs, rowptr, entbase, entk, coef, r, W)
        V[r, :no] = row[ng:]
        V[r, no] = row[:ng] @ g if ng else 0.0

    Pm, q = np.zeros((no, no)), np.zeros(no)
    gt = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    for a, b, n, pw, d, pa, flags, _ma, _mb, _pad in gt:
        w, aim = params[pw], params[pa]
        if flags & P.GT_FLAG_DIAG:       # rows coef_k e_{a+k}: diagonal update
            cf = dt[it[H["DOFF_DIAGCOEF"]] + b:it[H["DOFF_DIAGCOEF"]] + b + n]
            idx = np.arange(a, a + n)
            Pm[idx, idx] += (w * cf) * cf
            q[idx] += w * (cf * (0.0 - aim))
            continue
        scale = 0.5 if flags & P.GT_FLAG_HALF else 1.0
        A = w * V[a:a + n, :no]
        if flags & P.GT_FLAG_P:
            Pm += A.T @ V[b:b + n, :no]
        q += A.T @ (scale * (V[d:d + n, no] - aim))

    G, h = np.zeros((nc, no)), np.zeros(nc)
    lims = _section(it, "OFF_LIMIT", it[H["NLIMIT"]] * P.LM_WORDS).reshape(-1, P.LM_WORDS)
    lax = _section(it, "OFF_LAX", it[H["NLAX"]] * P.LX_WORDS).reshape(-1, P.LX_WORDS)
    rowlimit = _section(it, "OFF_ROWLIMIT", nc)
    for R in range(nc):
        lm = lims[rowlimit[R]]
        r, na = R - lm[0], lm[2]
        pick = lambda p0, rows, width: params[p0 + (0 if rows == 1 else r) * width:][:width]
        arrow, center = pick(lm[4], lm[5], na), pick(lm[6], lm[7], na)
        extreme = params[lm[8] + (0 if lm[9] == 1 else r)]
        ac = ad = 0.0
        for ax in range(na):
            off, rows = lax[lm[3] + ax]
            src_row = V[off + (0 if rows == 1 else r)]
            G[R] += arrow[ax] * src_row[:no]
            ac += arrow[ax] * center[ax]
            ad += arrow[ax] * src_row[no]
        h[R] = (extreme + ac) - ad

    pmrows, pm_nent = it[H["PMROWS"]], it[H["PM_NENT"]]
    prp = _section(it, "OFF_PM_ROWPTR", pmrows + 1)
    peb, pek = _section(it, "OFF_PM_ENTBASE", pm_nent), _section(it, "OFF_PM_ENTK", pm_nent)
    pcoef = dt[it[H["DOFF_PM_ENTCOEF"]]:it[H["DOFF_PM_ENTCOEF"]] + pm_nent]
    PM = np.stack([_row(plan, srcs, prp, peb, pek, pcoef, r, W) for r in range(pmrows)]) \
        if pmrows else np.zeros((0, W))
    if it[H["PM_NFD"]]:                   # the same matrices from the element program
        nfd, nops = int(it[H["PM_NFD"]]), int(it[H["PM_NOPS"]])
        pmap = _section(it, "OFF_PM_MAP", pmrows * W)
        fdp = _section(it, "OFF_PM_FDPTR", nfd + 1)
        ops = _section(it, "OFF_PM_OP", nops * 2).view(np.uint32).reshape(-1, 2)
        pool = dt[it[H["DOFF_PM_POOL"]]:it[H["DOFF_PM_POOL"]] + it[H["PM_NPOOL"]]]
        flat = [np.asarray(s_, dtype=np.float64).reshape(-1) for s_ in srcs]
        PM2 = np.zeros(pmrows * W)
        for e in range(pmrows * W):
            m = pmap[e]
            if m < 0:
                continue
            acc = 0.0
            for off, word in ops[fdp[m]:fdp[m + 1]]:
                sid, cid = int(word) & 255, int(word) >> 8
                acc += pool[cid] * (1.0 if sid == 255 else flat[sid][off])
            PM2[e] = acc
        assert np.allclose(PM2.reshape(pmrows, W), PM, rtol=1e-14, atol=0)
    return {"V": V, "P": Pm, "q": q, "G": G, "h": h, "PM": PM}


def run_fused_workspace(plan, given, sources=None):
    """Workspace V built by the *fused program* (per-element op lists over the
    on-chip source arena), for comparison with the row-set program's V."""
    it, dt = plan.itab, plan.dtab
    no, ldv = plan.no, plan.ldv
    srcs = [s.array for s in plan.sources] if sources is None else sources
    g = np.asarray(given, dtype=float).ravel()
    assert it[H["FUSED_OK"]] == 1
    arena = np.zeros(it[H["ARENA_TOTAL"]])
    arena[0] = 1.0
    rec = _section(it, "OFF_ARENA", 2 * it[H["NSRC"]]).reshape(-1, 2)
    for (off, size), a in zip(rec, srcs):
        assert a.size == size
        arena[off:off + size] = a.ravel()
    nfd, nops = it[H["NFD"]], it[H["NOPS"]]
    fd_idx = _section(it, "OFF_FD_IDX", nfd)
    fd_ptr = _section(it, "OFF_FD_PTR", nfd + 1)
    ops = _section(it, "OFF_OP", 2 * nops).view(np.uint32).reshape(-1, 2)
    pool = dt[it[H["DOFF_COEFPOOL"]]:it[H["DOFF_COEFPOOL"]] + it[H["NCOEF"]]]
    V = np.zeros(plan.rtot * ldv)
    for i in range(nfd):
        acc = 0.0
        for o in range(fd_ptr[i], fd_ptr[i + 1]):
            src, packed = int(ops[o, 0]), int(ops[o, 1])
            gi, cid = (packed >> 16) - 1, packed & 0xFFFF
            acc += pool[cid] * arena[src] * (g[gi] if gi >= 0 else 1.0)
        V[fd_idx[i]] = acc
    V = V.reshape(plan.rtot, ldv)[:, :no + 1]
    # tile masks promise exact zeros outside the marked 16-column tiles
    gt = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    for a, b, n, pw, d, pa, flags, ma, mb, _pad in gt:
        if flags & P.GT_FLAG_DIAG:
            continue
        for off, mask in ((a, ma), (b, mb)):
            if off < 0:
                continue
            for t in range((no + 15) // 16):
                if not (mask >> min(t, 30)) & 1:
                    assert not V[off:off + n, 16 * t:min(16 * t + 16, no)].any()
    return V


def run_resident(plan, given, params=None, sources=None):
    """P, q, G, h of one instance from the *resident program* (the tables of the persistent
    kernel, csrc/resident.hip): the input image assembled load by load, the per-thread
    compose ops (shared elements added up), the per-wavefront (tile, term) items with the
    gradient riding in column ``no`` of the B operand, the row records of G."""
    it, dt = plan.itab, plan.dtab
    ng, no, nc, ldv = plan.ng, plan.no, plan.nc, plan.ldv
    assert it[H["RS_OK"]] == 1
    srcs = [s.array for s in plan.sources] if sources is None else sources
    params = plan.params if params is None else np.asarray(params, dtype=float)
    g = np.asarray(given, dtype=float).ravel()
    nsrc, nparams = it[H["NSRC"]], it[H["NPARAMS"]]
    # ---- the image: one LDS-DMA load moves `unit` bytes per lane to consecutive addresses
    unit, nchunk, img = int(it[H["RS_UNIT"]]), int(it[H["RS_NCHUNK"]]), int(it[H["RS_IMG"]])
    const = dt[it[H["DOFF_RS_CONST"]]:it[H["DOFF_RS_CONST"]] + 4]
    streams = [np.ascontiguousarray(a, dtype=np.float64).ravel().view(np.uint8) for a in srcs]
    streams += [g.view(np.uint8), np.ascontiguousarray(params).view(np.uint8),
                np.ascontiguousarray(const).view(np.uint8)]
    meta = _section(it, "OFF_RS_INMETA", nchunk * 64 * 2).reshape(-1, 2)
    image = np.full(img * 8, 0xFF, dtype=np.uint8)          # NaN pattern where nothing lands
    assert meta.shape[0] * unit == it[H["RS_IMG_DMA"]] * 8 <= img * 8
    for lane, (st, off) in enumerate(meta):
        assert 0 <= st < nsrc + 3 and off % unit == 0 and off + unit <= streams[st].size
        image[lane * unit:(lane + 1) * unit] = streams[st][off:off + unit]
    image = image.view(np.float64)
    # horizon matrices generated on chip: tables (A^{k+1})[i][j] and (A^d B)[i][j], k, d < N,
    # from the group's A and B (which arrived through the streams of its first two sources)
    lti = _section(it, "OFF_RS_LTI", it[H["RS_NLTI"]] * P.RS_LTI_WORDS).reshape(-1, P.RS_LTI_WORDS)
    nab = int(it[H["RS_AB"]])                    # (A, B) arrive apart, by 4-byte loads
    abmeta = _section(it, "OFF_RS_ABMETA", nab * 4).reshape(-1, 2)
    ab = np.full(nab * 8, 0xFF, dtype=np.uint8)
    for lane, (st, off) in enumerate(abmeta):
        assert 0 <= st < nsrc + 3 and off % 4 == 0 and off + 4 <= streams[st].size
        ab[lane * 4:lane * 4 + 4] = streams[st][off:off + 4]
    ab = ab.view(np.float64)
    for n, m, N, ia, ib, ta, tb, tp in lti:
        Am = ab[ia:ia + n * n].reshape(n, n).copy()
        Bm = ab[ib:ib + n * m].reshape(n, m).copy()
        for k in range(N):
            image[ta + k * n * n:ta + (k + 1) * n * n] = np.linalg.matrix_power(Am, k + 1).ravel()
            image[tb + k * n * m:tb + (k + 1) * n * m] = (np.linalg.matrix_power(Am, k) @ Bm).ravel()
    prm = image[it[H["RS_IMG_PARAMS"]]:]
    assert image[0] == 1.0 and image[it[H["RS_IMG_GIVEN"]] + ng] == 1.0 and prm[nparams] == 0.0
    assert np.array_equal(prm[:nparams], params)
    # ---- compose
    jc = it[H["RS_JC"]]
    src = _section(it, "OFF_RS_SRC", jc * P.RS_NT).reshape(jc, P.RS_NT)
    gix = _section(it, "OFF_RS_GIDX", jc * P.RS_NT).reshape(jc, P.RS_NT)
    dst = _section(it, "OFF_RS_DST", jc * P.RS_NT).reshape(jc, P.RS_NT)
    cf = dt[it[H["DOFF_RS_COEF"]]:it[H["DOFF_RS_COEF"]] + jc * P.RS_NT].reshape(jc, P.RS_NT)
    assert plan.rtot % 4 == 0                           # row-sets are padded to groups of four
    # the kernel's layout (plan.py Workspace): row major, leading dimension RS_LDV, RS_VROW0 zero rows
    # in front, a row holds its window of the unknowns, d in column RS_VD and the one behind it
    geo = plan.workspace
    vldv, vd, vrow0 = int(it[H["RS_LDV"]]), int(it[H["RS_VD"]]), int(it[H["RS_VROW0"]])
    assert (vldv, vd, vrow0, geo.compact) == (geo.ldv, geo.vd, geo.row0, it[H["RS_COMPACT"]])
    assert geo.compact or (vldv, vd, vrow0) == (ldv, no, 0)
    V = np.zeros((vrow0 + plan.rtot) * vldv + 16)
    split = _section(it, "OFF_RS_SPLIT", it[H["RS_NSPLIT"]])
    written = np.zeros(V.size, dtype=np.int64)
    for t in range(P.RS_NT):
        acc = 0.0
        for j in range(jc):
            acc += cf[j, t] * image[src[j, t]] * image[gix[j, t]]
            d = int(dst[j, t])
            if d >= 0:
                if d & P.RS_DST_ACC:
                    d &= ~P.RS_DST_ACC
                    assert d in split
                    V[d] += acc
                    written[d] += 1
                else:
                    assert written[d] == 0
                    V[d] = acc
                    written[d] = 99
                acc = 0.0
    assert all(written[d] in (1, 2) for d in split)
    rowstart = (vrow0 + np.arange(plan.rtot)) * vldv
    assert not V[rowstart + vd + 1].any() and not V[:vrow0 * vldv].any()
    V[rowstart + vd + 1] = 1.0                                 # the ones column (set once per launch)
    assert ldv >= no + 2 and vldv % 4 == 2
    # the same workspace as the fused kernel keeps it: dense, columns = unknowns, d, -
    Vrc = np.zeros((plan.rtot, ldv))
    for r in range(plan.rtot):
        c0, w = int(geo.c0[r]), min(int(geo.w[r]), no - int(geo.c0[r]))
        Vrc[r, c0:c0 + w] = V[rowstart[r]:rowstart[r] + w]
        Vrc[r, no] = V[rowstart[r] + vd]
    # ---- Hessian and gradient: packs of four 4x4 blocks (plan_tables.h RT_*)
    nb = (no + 3) // 4
    TW = P.RS_TRIP_WORDS
    trips = _section(it, "OFF_RS_TRIP", (it[H["RS_NTRIP"]] + 2) * TW).reshape(-1, TW)
    assert not trips[-2:].any()                              # read ahead by the kernel
    trips = trips[:-2]
    wtrip = _section(it, "OFF_RS_WTRIP", P.RS_WAVES * 2).reshape(-1, 2)
    # diagonal gterms: the per-column tables (weight, aim slots and coefficients of at most two
    # terms; slot NPARAMS is the 0.0 behind the parameters), checked against the gterm records
    dpar = _section(it, "OFF_RS_DPAR", no * 4).reshape(no, 4)
    dcoef = dt[it[H["DOFF_RS_DCOEF"]]:it[H["DOFF_RS_DCOEF"]] + 2 * no].reshape(no, 2)
    prm0 = np.append(prm, 0.0)
    dP = (prm0[dpar[:, 0]] * dcoef[:, 0]) * dcoef[:, 0] + (prm0[dpar[:, 2]] * dcoef[:, 1]) * dcoef[:, 1]
    dq = (prm0[dpar[:, 0]] * (dcoef[:, 0] * (0.0 - prm0[dpar[:, 1]]))
          + prm0[dpar[:, 2]] * (dcoef[:, 1] * (0.0 - prm0[dpar[:, 3]])))
    dP_ref, dq_ref = np.zeros(no), np.zeros(no)
    gt = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    for a, b, n, pw, d, pa, flags, _ma, _mb, _pad in gt:
        if flags & P.GT_FLAG_DIAG:
            c = dt[it[H["DOFF_DIAGCOEF"]] + b:it[H["DOFF_DIAGCOEF"]] + b + n]
            idx = np.arange(a, a + n)
            dP_ref[idx] += (prm[pw] * c) * c
            dq_ref[idx] += prm[pw] * (c * (0.0 - prm[pa]))
    assert np.allclose(dP, dP_ref, rtol=1e-15, atol=0) and np.allclose(dq, dq_ref, rtol=1e-15, atol=0)
    Pm, q = np.full((no, no), np.nan), np.full(no, np.nan)
    four = np.arange(4)
    assert wtrip[:, 1].sum() == len(trips)
    rbytes = vldv * 8
    for first_trip, count in wtrip:
        acc, open_pack, S = None, None, np.zeros((4, 4, 4))
        for x in trips[first_trip:first_trip + count]:
            word = int(x[P.RT_WORD])
            rows, short = word & 31, (word >> P.RT_SHORT) & 1
            half, nop = (word >> P.RT_HALF) & 1, (word >> P.RT_NOP) & 1
            live, qmask = (word >> P.RT_LIVE) & 15, (word >> P.RT_QMASK) & 15
            pack = (int(x[P.RT_BI]), int(x[P.RT_BJ]), live, qmask)
            assert rows in (0, 4, 16) and live and not qmask & ~live and short == (rows == 4)
            if (word >> P.RT_FIRST) & 1:
                assert open_pack is None and not S.any()
                acc, open_pack = np.zeros((4, 4, 4)), pack
            assert open_pack == pack
            assert rows or ((word >> P.RT_FIRST) & 1 and (word >> P.RT_LAST) & 1)
            w, aim = prm[x[P.RT_W] // 8], prm[x[P.RT_AIM] // 8]
            # operands by address, as the kernel reads them: 32 bytes at block bi (bj) of each row
            # (compact: the offsets have the window's first column taken off)
            assert x[P.RT_A] % 32 == 0 and x[P.RT_B] % 32 == 0
            assert geo.compact or (x[P.RT_A] % (4 * rbytes) == 0 and x[P.RT_B] % (4 * rbytes) == 0)
            if rows:
                assert (x[P.RT_D] - vd * 8) % (4 * rbytes) == 0
            for g in range(4):
                bi, bj = (pack[0] >> (8 * g)) & 255, (pack[1] >> (8 * g)) & 255
                assert bi < nb and bj < nb
                for k in range(rows):
                    ia = int(x[P.RT_A]) // 8 + 4 * bi + k * vldv
                    assert ia >= 0
                    av = V[ia + four]
                    if (qmask >> g) & 1:          # B operand of a block of q: d, ones (, junk, junk)
                        idd = int(x[P.RT_D]) // 8 + k * vldv
                        bv = np.array([V[idd], V[idd + 1], 0.0, 0.0])
                    else:
                        ib = int(x[P.RT_B]) // 8 + 4 * bj + k * vldv
                        assert ib >= 0
                        bv = V[ib + four]
                    S[g] += np.outer(av, bv)
            if (word >> P.RT_TERM_END) & 1:       # the term's sum enters the pack with its weight
                assert rows
                ws = (0.5 * w) if half else w
                for g in range(4):
                    if (qmask >> g) & 1:
                        acc[g][:, 0] += ws * (S[g][:, 0] - aim * S[g][:, 1])
                    else:
                        acc[g] += (0.0 if nop else w) * S[g]
                S = np.zeros((4, 4, 4))
            if (word >> P.RT_LAST) & 1:
                for g in range(4):
                    if not (live >> g) & 1:
                        continue
                    bi, bj = (pack[0] >> (8 * g)) & 255, (pack[1] >> (8 * g)) & 255
                    for i in range(4):
                        row = 4 * bi + i
                        if row >= no:
                            continue
                        if (qmask >> g) & 1:
                            assert np.isnan(q[row])
                            q[row] = acc[g][i, 0] + dq[row]
                            continue
                        for jx in range(4):
                            col = 4 * bj + jx
                            if col < no:
                                assert np.isnan(Pm[row, col])
                                Pm[row, col] = acc[g][i, jx] + (dP[row] if row == col else 0.0)
                                if it[H["RS_SYM"]] and bi != bj:
                                    assert np.isnan(Pm[col, row])
                                    Pm[col, row] = Pm[row, col]
                open_pack = None
                assert not S.any()                       # every term was closed
        assert open_pack is None
    assert not np.isnan(q).any()
    Pm[np.isnan(Pm)] = 0.0        # blocks no term reaches: zeroed once per workgroup, never written
    # ---- constraint rows
    rr = _section(it, "OFF_RS_RR", nc * P.RS_RR_WORDS).reshape(nc, P.RS_RR_WORDS)
    rrwin = _section(it, "OFF_RS_RRWIN", nc if geo.compact else 0).view(np.uint32)
    G, h = np.zeros((nc, no)), np.zeros(nc)
    if it[H["RS_NGDESC"]]:                              # descriptors of the 16-byte pieces of G
        gd = _section(it, "OFF_RS_GDESC", it[H["RS_NGDESC"]] * 2).view(np.uint32).reshape(-1, 2)
        assert it[H["RR_PACKED"]] and len(gd) >= nc * (no // 2)
        gfix = _section(it, "OFF_RS_GFIX", 2 * P.RS_GDESC_THREADS if it[H["RS_NGFIX"]] else 0) \
            .view(np.uint32).reshape(-1, 2)
        fixed = {t + int(w >> 16) * P.RS_GDESC_THREADS for t, (w, _) in enumerate(gfix)
                 if w >> 16 != P.RS_GFIX_NONE}
        assert len(fixed) == it[H["RS_NGFIX"]]
        for e in range(nc * (no // 2)):
            R, cp = divmod(e, no // 2)
            v0, v1, a0, a1 = rr[R, 0] + 2 * cp, rr[R, 1] + 2 * cp, rr[R, 4], rr[R, 5]
            as_is = gd[e, 0] == v0 | (v1 << 16) and gd[e, 1] == a0 | (a1 << 16)
            swapped = gd[e, 0] == v1 | (v0 << 16) and gd[e, 1] == a1 | (a0 << 16)
            assert as_is or swapped or geo.compact       # (compact: the piece inside the axis' window)
            # a round marked "one axis": the second of the descriptor is structurally zero here
            u, w = divmod(e // 64, P.RS_GDESC_THREADS // 64)
            if (int(it[H["RS_GSINGLE"]]) >> (u * (P.RS_GDESC_THREADS // 64) + w)) & 1 and e not in fixed:
                second, arrow = int(gd[e, 0] >> 16), int(gd[e, 1] >> 16)
                # ... or absent (the always-zero parameter slot)
                assert (V[second] == 0.0 and V[second + 1] == 0.0) or arrow == nparams
    Gdesc = None
    if it[H["RS_NGDESC"]]:
        # ... and G as the descriptor path writes it: rounds of one-axis pieces read the first axis
        # only; the thread that owns a piece on the list RS_GFIX adds that piece's second axis
        T, U = P.RS_GDESC_THREADS, P.RS_GDESC_PIECES
        pieces = nc * (no // 2)
        Gflat = np.full(pieces * 2, np.nan)
        for e in range(pieces):
            u, t = divmod(e, T)
            d = gd[e]
            v0, v1, a0, a1 = int(d[0] & 0xFFFF), int(d[0] >> 16), int(d[1] & 0xFFFF), int(d[1] >> 16)
            val = prm[a0] * V[v0:v0 + 2]
            if not (int(it[H["RS_GSINGLE"]]) >> (u * (T // 64) + t // 64)) & 1:
                val = prm[a1] * V[v1:v1 + 2] + val
            elif len(gfix) and gfix[t, 0] >> 16 == u:
                assert (int(gfix[t, 0] & 0xFFFF), int(gfix[t, 1])) == (v1, a1)
                val = prm[a1] * V[v1:v1 + 2] + val
            Gflat[2 * e:2 * e + 2] = val
        for t, (w, arrow) in enumerate(gfix):
            assert w >> 16 == P.RS_GFIX_NONE and (w & 0xFFFF, arrow) == (0, nparams) \
                or t + int(w >> 16) * T < pieces
        Gdesc = Gflat.reshape(nc, no)
    for R in range(nc):
        rec = rr[R]
        ac = ad = 0.0
        for ax in range(rec[12]):
            arrow = prm[rec[4 + ax]]
            assert rec[ax] % vldv == 0                       # the first stored column of a row
            if geo.compact:                                   # ... which holds its window only
                c0 = 2 * ((int(rrwin[R]) >> (16 * ax)) & 255)
                w = min(2 * ((int(rrwin[R]) >> (16 * ax + 8)) & 255), no - c0)
                row = rec[ax] // vldv - vrow0
                assert (c0, 2 * ((int(rrwin[R]) >> (16 * ax + 8)) & 255)) == (geo.c0[row], geo.w[row])
            else:
                c0, w = 0, no
            G[R, c0:c0 + w] += arrow * V[rec[ax]:rec[ax] + w]
            ac += arrow * prm[rec[8 + ax]]
            ad += arrow * V[rec[ax] + vd]
        for ax in range(rec[12], P.RS_AXMAX):       # the kernel's fast path reads two axes
            assert prm[rec[4 + ax]] == 0.0
        h[R] = (prm[rec[13]] + ac) - ad
    if Gdesc is not None:
        assert np.allclose(Gdesc, G, rtol=1e-15, atol=0)
        G = Gdesc
    out = {"P": Pm, "q": q, "G": G, "h": h}
    if it[H["CSC_PNNZ"]] or it[H["CSC_GNNZ"]]:      # the CSC hand-off: data arrays as the kernel writes them
        ldp = no + (no & 1)
        Pl = np.zeros((no, ldp))
        Pl[:, :no] = Pm
        cp = _section(it, "OFF_CSC_P", it[H["CSC_PNNZ"]])
        assert ((cp // ldp < no) & (cp % ldp < no)).all()
        out["P_data"] = Pl.ravel()[cp]
        cg = _section(it, "OFF_CSC_G", it[H["CSC_GNNZ"]] * 2).view(np.uint32).reshape(-1, 2)
        v0, v1 = (cg[:, 0] & 0xFFFF).astype(int), (cg[:, 0] >> 16).astype(int)
        a0, a1 = (cg[:, 1] & 0xFFFF).astype(int), (cg[:, 1] >> 16).astype(int)
        if it[H["CSC_GSINGLE"]]:                   # the second axis is never read: it must not matter
            assert ((V[v1] == 0.0) | (a1 == nparams)).all()
            out["G_data"] = prm[a0] * V[v0]
        else:
            out["G_data"] = prm[a1] * V[v1] + prm[a0] * V[v0]
    out["V"] = Vrc
    return out


def _tiled_streams(plan, srcs, ab):
    """The streams the column tables address: the launch's sources -- with the horizon tables of
    every generated group in the places of its U_j (TB, plan_tables.h TL_*) and of its S (TA),
    built from the group's (A, B) by the reference's recurrence -- and, last, the plan's dtab."""
    it, dt = plan.itab, plan.dtab
    streams = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in srcs]
    streams += [None] * (P.T_SID_CONST - len(streams)) + [dt]
    lti = _section(it, "OFF_T_LTI", it[H["T_NLTI"]] * P.T_LTI_WORDS).reshape(-1, P.T_LTI_WORDS)
    for g, (n, m, N, ids0, _ta, _tb, _p0, _p1) in enumerate(lti):
        ids = _section(it, "OFF_T_LTI_IDS", ids0 + m + 1)[ids0:]
        Am, Bm = (np.asarray(x, dtype=float) for x in ab[g])
        TA = np.zeros((N, n, n))
        TB = np.zeros((n, m, 2 * N))
        X = np.hstack([Bm, Am])                      # X_d = A^d [B | A]
        for d in range(N):
            TB[:, :, N + d] = X[:, :m]
            TA[d] = X[:, m:].T                       # S[k][j][i] = (A^{k+1})[i][j]
            X = Am @ X
        for j in range(m):
            streams[ids[j]] = TB.ravel()
        streams[ids[m]] = TA.ravel()
    return streams


def run_preview_tables(plan, given, optim, sources=None, ab=None):
    """Rows of every definition, ``Mg @ given + Mo @ optim``, as csrc/preview.hip's staged kernel computes
    them from the plan's unrolled tables (plan_tables.h H_T_NP1): y[t] = sum over the base row's entries
    of stream[offset] * [given ; optim][column]; row r = sum of coef * y[base row]."""
    it, dt = plan.itab, plan.dtab
    assert it[H["T_NP1"]] >= 0
    srcs = [s.array for s in plan.sources] if sources is None else sources
    streams = _tiled_streams(plan, srcs, ab or [])
    x = np.concatenate([np.asarray(given, dtype=float).ravel(), np.asarray(optim, dtype=float).ravel()])
    nbrow = int(_section(it, "OFF_T_BROW0", it[H["NBASE"]] + 1)[-1])
    ptr = _section(it, "OFF_T_P1PTR", nbrow + 1)
    ent = _section(it, "OFF_T_P1ENT", it[H["T_NP1"]] * 2).view(np.uint32).reshape(-1, 2)
    assert ptr[0] == 0 and ptr[-1] == len(ent) and (np.diff(ptr) >= 0).all()
    y = np.zeros(nbrow)
    for t in range(nbrow):
        for off, tag in ent[ptr[t]:ptr[t + 1]]:
            y[t] += streams[int(tag) >> 24][int(off)] * x[int(tag) & 0xFFFFFF]
    rowptr = _section(it, "OFF_PM_ROWPTR", it[H["PMROWS"]] + 1)
    p2y = _section(it, "OFF_T_P2Y", it[H["PM_NENT"]])
    coef = dt[it[H["DOFF_PM_ENTCOEF"]]:it[H["DOFF_PM_ENTCOEF"]] + it[H["PM_NENT"]]]
    return np.array([np.dot(coef[rowptr[r]:rowptr[r + 1]], y[p2y[rowptr[r]:rowptr[r + 1]]])
                     for r in range(it[H["PMROWS"]])])


def _sext24(x):
    x = np.asarray(x, dtype=np.int64) & 0xFFFFFF
    return np.where(x >= 1 << 23, x - (1 << 24), x)


def run_tiled(plan, given, params=None, sources=None, ab=None):
    """P, q, G, h of one instance from the tables of the tiled kernel (csrc/tiled.hip): rows
    composed through the column tables, the Hessian accumulated stage by stage on the 16-column
    tiles the stage's masks admit (a mask that hides a non-zero tile shows as a wrong P), the
    gradient from the same weighted rows and d = Mg . given, G and h from the row records."""
    it, dt = plan.itab, plan.dtab
    ng, no, nc, ldv = plan.ng, plan.no, plan.nc, plan.ldv
    assert it[H["T_CI_OK"]] == 1
    srcs = [s.array for s in plan.sources] if sources is None else sources
    params = plan.params if params is None else np.asarray(params, dtype=float)
    g = np.asarray(given, dtype=float).ravel()
    streams = _tiled_streams(plan, srcs, ab)
    nbase, nop = int(it[H["NBASE"]]), int(it[H["T_NOP"]])
    cig = _section(it, "OFF_T_CIG", nbase * ng * 2).view(np.uint32).astype(np.int64).reshape(nbase, ng, 2)
    cio = _section(it, "OFF_T_CIO", nbase * nop * 2).view(np.uint32).astype(np.int64).reshape(nbase, nop, 2)
    assert it[H["OFF_T_CIO"]] % 4 == 0
    rtot, nent = int(it[H["RTOT"]]), int(it[H["NENT"]])
    rowptr = _section(it, "OFF_ROWPTR", rtot + 1)
    entbase, entk = _section(it, "OFF_ENTBASE", nent), _section(it, "OFF_ENTK", nent)
    coef = dt[it[H["DOFF_ENTCOEF"]]:it[H["DOFF_ENTCOEF"]] + nent]

    def base_row(ci, u, k):
        off, meta = ci[u, :, 0], ci[u, :, 1]
        sid, rs = meta >> 24, _sext24(meta)
        out = np.zeros(off.size)
        for s in np.unique(sid):
            pick = sid == s
            out[pick] = streams[s][off[pick] + k * rs[pick]]
        return out

    def row(ci, r):
        out = np.zeros(ci.shape[1])
        for e in range(rowptr[r], rowptr[r + 1]):
            out += coef[e] * base_row(ci, entbase[e], entk[e])
        return out

    Vo = np.stack([row(cio, r) for r in range(rtot)]) if rtot else np.zeros((0, nop))
    assert not Vo[:, no:].any()                                  # the pad columns read 0.0
    d = np.asarray([row(cig, r) @ g for r in range(rtot)]) if ng else np.zeros(rtot)

    Pm, q = np.zeros((nop, nop)), np.zeros(nop)
    nblk = nop // P.T_BLOCK
    nstage = int(it[H["T_NSTAGE"]])
    stages = _section(it, "OFF_T_STAGE", nstage * P.T_STAGE_WORDS).view(np.uint32) \
        .astype(np.int64).reshape(-1, P.T_STAGE_WORDS)
    srow = _section(it, "OFF_T_SROW", nstage * 32).reshape(nstage, 2, 16)
    scoef = dt[it[H["T_DOFF_SCOEF"]]:it[H["T_DOFF_SCOEF"]] + nstage * 32].reshape(nstage, 2, 16)
    pig = _section(it, "OFF_T_PIG", nstage * 16 * P.T_PIG_MAX * 2).reshape(nstage, 16, P.T_PIG_MAX, 2)
    G, h = np.full((nc, no), np.nan), np.zeros(nc)
    prm0 = np.append(params, 0.0)

    def blockbits(m, b):                                  # the 8 tiles of block b (fold onto bit 63)
        return [(m >> min(8 * b + j, 63)) & 1 for j in range(8)]

    last_cls = 0
    nlti = int(it[H["T_NLTI"]])
    toeplitz_plan = bool(it[H["T_TOEPLITZ"]])
    for sx, (arow, brow, drow, info, pw, pa, mal, mah, mbl, mbh, base, _, ua, ub, sba, sbb) in enumerate(stages):
        n, fl, cls = info & 255, (info >> 8) & 255, info >> 16
        assert not toeplitz_plan or fl & P.TS_FLAG_TOEPLITZ
        if fl & P.TS_FLAG_TOEPLITZ:
            # the window of the group's Toeplitz table the kernel reads instead of composing rows:
            # element (row r, column c) = coef * TB[offset(c) - sboff + U + r] where column c's table
            # entry is one of the group's U_j with one element per base row, else 0
            assert nlti == 1
            lti = _section(it, "OFF_T_LTI", P.T_LTI_WORDS)
            ids = set(_section(it, "OFF_T_LTI_IDS", lti[3] + lti[1] + 1)[lti[3]:lti[3] + lti[1]])
            for side, (row0, U, sb) in enumerate(((arow, ua, sba), (brow, ub, sbb))):
                if side == 1 and not fl & P.TS_FLAG_P:
                    continue
                b_ = (base & 0xFFFF) if side == 0 else (base >> 16)
                off, meta = cio[b_, :, 0], cio[b_, :, 1]
                sid, rs = meta >> 24, _sext24(meta)
                valid = np.isin(sid, list(ids))
                assert np.all(rs[valid] == 1) and np.all(sid[~valid] == P.T_SID_CONST) and np.all(rs[~valid] == 0)
                tb = streams[next(iter(ids))]
                win = np.zeros((n, nop))
                for r in range(n):
                    win[r, valid] = scoef[sx, side, 0] * tb[off[valid] - sb + U + r]
                assert np.array_equal(win, Vo[row0:row0 + n]), "Toeplitz window"
        assert cls >= last_cls and (cls > 0) == bool(fl & P.TS_FLAG_P)   # sorted by class
        last_cls = cls
        ma, mb = mal | (mah << 32), mbl | (mbh << 32)
        w, aim = params[pw], params[pa]
        scale = 0.5 if fl & P.TS_FLAG_HALF else 1.0
        assert bool(fl & P.TS_FLAG_SAME) == (arow == brow)

        def rows(side, row0, flag, ci=cio):               # the stage's rows as the kernel composes them
            if fl & flag:
                b_ = (base & 0xFFFF) if side == 0 else (base >> 16)
                out = np.stack([scoef[sx, side, i] * base_row(ci, b_, srow[sx, side, i]) for i in range(n)])
                assert not scoef[sx, side, n:].any()
                return out
            return np.stack([row(ci, r) for r in range(row0, row0 + n)])

        Au = rows(0, arow, P.TS_FLAG_SIMPLE_A)
        assert np.array_equal(Au, Vo[arow:arow + n])
        B = rows(1, brow, P.TS_FLAG_SIMPLE_B)
        assert np.array_equal(B, Vo[brow:brow + n])
        A = w * Au
        r = scale * (d[drow:drow + n] - aim)
        for bi in range(nblk):
            ca = slice(bi * P.T_BLOCK, (bi + 1) * P.T_BLOCK)
            hit_a = any(blockbits(ma, bi))
            if hit_a:                                      # (diagonal workgroup of block bi)
                q[ca] += A[:, ca].T @ r
            else:
                assert not A[:, ca].any()
            if hit_a or fl & P.TS_FLAG_G:                  # rows of G riding on the A rows
                for i in range(n):
                    for R, slot in pig[sx, i]:
                        if R >= 0:
                            assert np.isnan(G[R, ca.start:min(ca.stop, no)]).all()    # written once
                            G[R, ca.start:min(ca.stop, no)] = (prm0[slot] * Au[i, ca])[:min(ca.stop, no) - ca.start]
            if not (fl & P.TS_FLAG_P) or not hit_a:
                continue
            for bj in range(nblk):
                if not any(blockbits(mb, bj)):
                    continue
                for ta in range(8):
                    for tb in range(8):
                        if ta % 4 < cls and tb % 4 < cls:
                            ra = slice(bi * P.T_BLOCK + 16 * ta, bi * P.T_BLOCK + 16 * ta + 16)
                            cb = slice(bj * P.T_BLOCK + 16 * tb, bj * P.T_BLOCK + 16 * tb + 16)
                            Pm[ra, cb] += A[:, ra].T @ B[:, cb]
        # what the masks and the class promise: nothing outside the multiplied tiles
        for t in range(nop // 16):
            if not ((ma >> min(t, 63)) & 1 and t % 4 < max(cls, 1 if not fl & P.TS_FLAG_P else cls)):
                assert not Au[:, 16 * t:16 * t + 16].any() or not (fl & P.TS_FLAG_P)
            if fl & P.TS_FLAG_P and not ((mb >> min(t, 63)) & 1 and t % 4 < cls):
                assert not B[:, 16 * t:16 * t + 16].any()
    Pm, q = Pm[:no, :no], q[:no]
    gt = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    want = 0
    for a, b, n, pw, dd, pa, flags, _ma, _mb, _pad in gt:
        if flags & P.GT_FLAG_DIAG:
            cf = dt[it[H["DOFF_DIAGCOEF"]] + b:it[H["DOFF_DIAGCOEF"]] + b + n]
            idx = np.arange(a, a + n)
            Pm[idx, idx] += (params[pw] * cf) * cf
            q[idx] += params[pw] * (cf * (0.0 - params[pa]))
        else:
            want += -(-n // 16)
    assert want == nstage

    rr = _section(it, "OFF_RS_RR", nc * P.RS_RR_WORDS).reshape(nc, P.RS_RR_WORDS)
    grow = _section(it, "OFF_T_GROW", nc * P.RS_AXMAX).reshape(nc, P.RS_AXMAX)
    grest = _section(it, "OFF_T_GREST", it[H["T_NGREST"]])
    for R in range(nc):
        x = rr[R]
        ac = ad = 0.0
        row_sum = np.zeros(no)
        for ax in range(x[12]):
            assert x[ax] == plan.workspace.rowstart(grow[R, ax])   # (the persistent kernel's index)
            r = grow[R, ax]
            row_sum += prm0[x[4 + ax]] * Vo[r, :no]
            ac += prm0[x[4 + ax]] * prm0[x[8 + ax]]
            ad += prm0[x[4 + ax]] * d[r]
        assert (grow[R, x[12]:] == -1).all()
        h[R] = (prm0[x[13]] + ac) - ad
        if R in grest:
            assert np.isnan(G[R]).all()
            G[R] = row_sum
        else:
            assert np.array_equal(G[R], row_sum)              # the riding rows: the same numbers
    assert not np.isnan(G).any()
    return {"P": Pm, "q": q, "G": G, "h": h, "d": d, "Vo": Vo[:, :no]}


def run_scan(plan, given, params=None, ab=None):
    """P, q, G, h of one instance as the tiled kernel's *scan form* computes them (csrc/tiled.hip
    toeplitz_scan_kernel, plan_tables.h T_SCAN*): the Hessian block of two input column blocks by
    the recurrence P[(j,l)][(j',l')] = C + P[(j,l+1)][(j',l'+1)] with the rank-K term C from the
    LAST rows of the states, rows of G straight out of the group's Toeplitz table, the gradient by
    correlating the table with s (d - aim), everything else (d, h, the diagonal terms, the rows of G
    that are no single state row) as the other forms do."""
    it, dt = plan.itab, plan.dtab
    K = int(it[H["T_SCAN"]])
    assert K > 0, "the plan has no scan form"
    ng, no, nc = plan.ng, plan.no, plan.nc
    params = plan.params if params is None else np.asarray(params, dtype=float)
    prm0 = np.append(params, 0.0)
    ref = run_tiled(plan, given, params=params, ab=ab)            # d, and what must come out
    d = ref["d"]
    lti = _section(it, "OFF_T_LTI", P.T_LTI_WORDS)
    n, m, N = int(lti[0]), int(lti[1]), int(lti[2])
    ids = _section(it, "OFF_T_LTI_IDS", lti[3] + m + 1)[lti[3]:]
    tb = _tiled_streams(plan, [s.array for s in plan.sources], ab)[ids[0]]
    assert tb.size == n * m * 2 * N
    tb = np.ascontiguousarray(tb.reshape(n * m, 2 * N)[:, N:]).ravel()     # Tc: without the zero halves

    def at(i):                       # Tc[i], 0 where the window leaves the row towards negative steps
        return tb[np.maximum(i, 0)]
    nblk = int(it[H["T_SCAN_NBLK"]])
    blk = _section(it, "OFF_T_SCAN_BLK", nblk * 2).reshape(nblk, 2)
    gt = _section(it, "OFF_T_SCAN_GT", K * P.T_SCAN_GT_WORDS).reshape(K, P.T_SCAN_GT_WORDS)
    gc = dt[it[H["T_DOFF_SCAN_GC"]]:it[H["T_DOFF_SCAN_GC"]] + K]
    colblk = _section(it, "OFF_T_SCAN_COLBLK", no)
    assert int((colblk < 0).sum()) == it[H["T_SCAN_NOTHER"]]
    for bx, (c0, pbase) in enumerate(blk):
        assert (colblk[c0:c0 + N] == bx).all() and pbase % N == 0
    lane = np.arange(N)

    def last_row(g, bx):            # M_g[N-1][columns of block bx] / c_g: TB[sboff + N-1 + pbase - l]
        return tb[gt[g, 0] + N - 1 + blk[bx, 1] - lane]

    Pm = np.zeros((no, no))
    for bi in range(nblk):
        for bj in range(nblk):
            Tl = np.stack([(params[gt[g, 1]] * gc[g]) * gc[g] * last_row(g, bj) for g in range(K)])
            run = np.zeros(N)
            for l in range(N - 1, -1, -1):
                beta = np.array([tb[gt[g, 0] + N - 1 + blk[bi, 1] - l] for g in range(K)])
                C = np.zeros(N)
                for g in range(K):                       # (the kernel's order of the K products)
                    C = C + beta[g] * Tl[g]
                carry = np.append(run[1:], 0.0)          # lane <- lane + 1, nothing behind the block
                run = C + carry
                Pm[blk[bi, 0] + l, blk[bj, 0]:blk[bj, 0] + N] = run
    q = np.zeros(no)
    for bx in range(nblk):
        acc = np.zeros(N)
        for g in range(K):
            w, aim, drow = params[gt[g, 1]], params[gt[g, 2]], gt[g, 3]
            for k in range(N):
                r = w * (d[drow + k] - aim)                  # (d carries the term's coefficient already)
                acc += r * (gc[g] * np.where(lane <= k, at(gt[g, 0] + k + blk[bx, 1] - lane), 0.0))
        q[blk[bx, 0]:blk[bx, 0] + N] = acc
    # ... and as the kernel does it: lam_l = rho_l + A^T lam_{l+1}, rho_l[i] = sum over the terms on
    # state i of (w c)(d[l] - aim), q[(j, l)] = B[:, j] . lam_l -- O(N n^2) instead of a correlation
    Am, Bm = (np.asarray(x, dtype=float) for x in ab[0])
    lam, q_rec = np.zeros(n), np.zeros(no)
    for l in range(N - 1, -1, -1):
        rho = np.zeros(n)
        for g in range(K):
            rho[gt[g, 0] // (m * N)] += (params[gt[g, 1]] * gc[g]) * (d[gt[g, 3] + l] - params[gt[g, 2]])
        lam = rho + Am.T @ lam
        for bx in range(nblk):
            q_rec[blk[bx, 0] + l] = Bm[:, blk[bx, 1] // N] @ lam
    assert np.abs(q_rec - q).max() <= 1e-12 * max(np.abs(q).max(), 1e-300), "gradient by the adjoint recursion"
    q = q_rec
    gtab = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    for a, b, nn, pw, dd, pa, flags, _ma, _mb, _pad in gtab:
        if flags & P.GT_FLAG_DIAG:
            cf = dt[it[H["DOFF_DIAGCOEF"]] + b:it[H["DOFF_DIAGCOEF"]] + b + nn]
            idx = np.arange(a, a + nn)
            Pm[idx, idx] += (params[pw] * cf) * cf
            q[idx] += params[pw] * (cf * (0.0 - params[pa]))
    grow = _section(it, "OFF_T_SCAN_GROW", nc * 2).reshape(nc, 2)
    gcoef = dt[it[H["T_DOFF_SCAN_GCOEF"]]:it[H["T_DOFF_SCAN_GCOEF"]] + nc]
    grest = _section(it, "OFF_T_SCAN_GREST", it[H["T_SCAN_NGREST"]])
    G = np.zeros((nc, no))
    for R in range(nc):
        if grow[R, 0] < 0:
            assert R in grest
            G[R] = ref["G"][R]                           # (composed through the column tables)
            continue
        assert R not in grest
        for bx in range(nblk):
            k = grow[R, 0] % N                           # (the row's step: columns l > k hold zeros)
            G[R, blk[bx, 0]:blk[bx, 0] + N] = prm0[grow[R, 1]] * (
                gcoef[R] * np.where(lane <= k, at(grow[R, 0] + blk[bx, 1] - lane), 0.0))
    return {"P": Pm, "q": q, "G": G, "h": ref["h"], "ref": ref}


def run_sweep(plan, given, A_steps, B_steps, params=None):
    """P, q, G, h of one instance as the sweep kernel computes them (csrc/sweep.hip, plan_tables.h SW_*)
    from per-step ``A_steps (N, n, n)``, ``B_steps (N, n, m)``: the free response of every axis, the
    backward recursions Psi_l = W_l + A_{l+1}^T Psi_{l+1} A_{l+1} and lam_l = rho_l + A_{l+1}^T lam_{l+1},
    the forward sweep of u = Phi(l, l'+1) B_l' per column (the rows of G of step l and P at and below the
    diagonal), the backward sweep of z = Phi(l', l+1)^T Psi_l' B_l' (P above the diagonal).  No horizon
    matrix is formed."""
    it, dt = plan.itab, plan.dtab
    assert it[H["SW_OK"]] == 1
    n, m, N, naxes = (int(it[H[k]]) for k in ("SW_N", "SW_M", "SW_HORIZON", "SW_NAXES"))
    ng, no, nc = plan.ng, plan.no, plan.nc
    params = plan.params if params is None else np.asarray(params, dtype=float)
    prm = np.append(params, 0.0)
    A, B = np.asarray(A_steps, dtype=float), np.asarray(B_steps, dtype=float)
    assert A.shape == (N, n, n) and B.shape == (N, n, m)
    g = np.asarray(given, dtype=float).ravel()
    axis = _section(it, "OFF_SW_AXIS", naxes * P.SW_AXIS_WORDS).reshape(naxes, P.SW_AXIS_WORDS)
    terms = _section(it, "OFF_SW_TERM", it[H["SW_NTERM"]] * P.SW_TERM_WORDS).reshape(-1, P.SW_TERM_WORDS)
    lw = P.SW_LIM_WORDS + P.SW_AXMAX * P.SW_LAX_WORDS
    lims = _section(it, "OFF_SW_LIM", it[H["SW_NLIM"]] * lw).reshape(-1, lw)
    col = _section(it, "OFF_SW_COL", no)
    cvec = dt[it[H["SW_DOFF_CVEC"]]:it[H["SW_DOFF_CVEC"]] + it[H["SW_NCVEC"]] * P.SW_NMAX].reshape(-1, P.SW_NMAX)

    def cv(off):
        return cvec[off // P.SW_NMAX, :n]

    # the free response x_k = A_k x_{k-1}, x_{-1} = the axis' initial state
    xbar = np.zeros((naxes, N, n))
    for a in range(naxes):
        x = g[axis[a, 0]:axis[a, 0] + n]
        for k in range(N):
            x = A[k] @ x
            xbar[a, k] = x
    # backward: Psi, lam -> gv[a][j][l] = Psi_l B_l[:, j], q
    W, rho = np.zeros((naxes, N, n, n)), np.zeros((naxes, N, n))
    for a, k0, ks, cnt, pw, pa, co, _ in terms:
        c = cv(co)
        for i in range(cnt):
            k = k0 + i * ks
            W[a, k] += prm[pw] * np.outer(c, c)
            rho[a, k] += (prm[pw] * (c @ xbar[a, k] - prm[pa])) * c
    gv, q = np.zeros((naxes, m, N, n)), np.zeros(no)
    for a in range(naxes):
        Psi, lam = np.zeros((n, n)), np.zeros(n)
        for l in range(N - 1, -1, -1):
            if l + 1 < N:
                Psi, lam = A[l + 1].T @ Psi @ A[l + 1], A[l + 1].T @ lam
            Psi, lam = Psi + W[a, l], lam + rho[a, l]
            for j in range(m):
                gv[a, j, l] = Psi @ B[l][:, j]
                q[axis[a, 1 + j] + l] = B[l][:, j] @ lam
    ca, cj, cl = col & 255, (col >> 8) & 255, col >> 16
    for c in range(no):
        assert axis[ca[c], 1 + cj[c]] + cl[c] == c
    Pm, G, h = np.full((no, no), np.nan), np.full((nc, no), np.nan), np.zeros(nc)
    # forward sweep
    u = np.zeros((no, n))
    for l in range(N):
        for c in range(no):
            if cl[c] == l:
                u[c] = B[l][:, cj[c]]
            elif cl[c] < l:
                u[c] = A[l] @ u[c]
        for a in range(naxes):
            for j in range(m):
                r = axis[a, 1 + j] + l
                for c in range(no):
                    if ca[c] != a:
                        Pm[r, c] = 0.0
                    elif cl[c] <= l:
                        Pm[r, c] = gv[a, j, l] @ u[c]
        for rec in lims:
            out0, cnt, nax, pe, pes = rec[:5]
            for i in range(cnt):
                row = np.zeros(no)
                hit = False
                for ax in range(nax):
                    a, k0, ks, co, parr, pas, pcen, pcs = rec[P.SW_LIM_WORDS + ax * P.SW_LAX_WORDS:][:8]
                    if k0 + i * ks != l:
                        continue
                    hit = True
                    row += np.where(ca == a, prm[parr + i * pas] * (u @ cv(co)), 0.0)
                if hit:
                    assert np.isnan(G[out0 + i]).all()             # (every axis of a line at the same step)
                    G[out0 + i] = row
    # backward sweep: above the diagonal
    z = np.zeros((no, n))
    for l in range(N - 1, -1, -1):
        for c in range(no):
            if cl[c] == l:
                z[c] = gv[ca[c], cj[c], l]
            elif cl[c] > l:
                z[c] = A[l + 1].T @ z[c]
        for a in range(naxes):
            for j in range(m):
                r = axis[a, 1 + j] + l
                for c in range(no):
                    if ca[c] == a and cl[c] > l:
                        assert np.isnan(Pm[r, c])
                        Pm[r, c] = B[l][:, j] @ z[c]
    assert not np.isnan(Pm).any() and not np.isnan(G).any()
    for rec in lims:
        out0, cnt, nax, pe, pes = rec[:5]
        for i in range(cnt):
            ac = ad = 0.0
            for ax in range(nax):
                a, k0, ks, co, parr, pas, pcen, pcs = rec[P.SW_LIM_WORDS + ax * P.SW_LAX_WORDS:][:8]
                ar = prm[parr + i * pas]
                ac += ar * prm[pcen + i * pcs]
                ad += ar * (cv(co) @ xbar[a, k0 + i * ks])
            h[out0 + i] = (prm[pe + i * pes] + ac) - ad
    gtab = _section(it, "OFF_GTERM", it[H["NGTERM"]] * P.GT_WORDS).reshape(-1, P.GT_WORDS)
    for a, b, nn, pw, dd, pa, flags, _ma, _mb, _pad in gtab:
        if flags & P.GT_FLAG_DIAG:
            cf = dt[it[H["DOFF_DIAGCOEF"]] + b:it[H["DOFF_DIAGCOEF"]] + b + nn]
            idx = np.arange(a, a + nn)
            Pm[idx, idx] += (params[pw] * cf) * cf
            q[idx] += params[pw] * (cf * (0.0 - params[pa]))
    return {"P": Pm, "q": q, "G": G, "h": h}
