"""GPU: the BASELINE configurations at their FULL per-GPU sizes, checked
through properties that need no CPU recomputation of the whole batch -- symmetry, equality
across instances where the inputs say so, affinity in the given vector, the LTV recurrence --
plus a few instances against the oracle; the LTV fill feeding an assembly end to end; and the
on-chip horizon tables of a marginally unstable, badly conditioned system.
Tolerance: 1e-10 relative (north star) where stated, else the tight regression bound."""
import numpy as np
import pytest

from helpers import RTOL, RTOL_TIGHT, assert_close
from mpcasm import problems
from oracle import qp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def _rel(t, ref):
    scale = ref.abs().max().item()
    return (t - ref).abs().max().item() / (scale if scale > 0 else 1.0)


def test_c3_lipm3d_at_16384_instances(gpu_api, torch_gpu):
    """C3: 3-D LIPM, N=32, no=96, nc=196, B=16384 on one GPU."""
    torch = torch_gpu
    from mpcasm.engine import Assembler

    form = problems.lipm3d(gpu_api, N=32)
    B = 16384
    rng = np.random.default_rng(20261)
    asm = Assembler(form, batch=B)
    ga = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    gb = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    P, q, G, h = (t.clone() for t in asm.assemble(ga))
    assert P.shape == (B, 96, 96) and G.shape == (B, 196, 96)
    # one model, one set of weights for the whole batch: P and G cannot differ between instances
    assert torch.equal(P, P[:1].expand_as(P)) and torch.equal(G, G[:1].expand_as(G))
    # every cost of C3 is a plain quadratic: P is symmetric
    assert _rel(P[0].T, P[0]) <= 1e-13
    # q and h are affine in the given vector
    qb, hb = (t.clone() for t in asm.assemble(gb)[1::2])
    q0, h0 = (t.clone() for t in asm.assemble(torch.zeros_like(ga))[1::2])
    qs, hs = asm.assemble(ga + 2.0 * gb)[1::2]
    assert _rel(qs, q + 2.0 * qb - 2.0 * q0) <= 1e-11
    assert _rel(hs, h + 2.0 * hb - 2.0 * h0) <= 1e-11
    for b in (0, 8191, B - 1):
        Ao, ho, Qo, qo = orc.assemble(form, ga[b].cpu().numpy().reshape(-1, 1))
        assert_close(P[b].cpu().numpy(), Qo, RTOL_TIGHT)
        assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
        assert_close(G[b].cpu().numpy(), Ao, RTOL_TIGHT)
        assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)


def test_c4_random_lti_at_1024_instances(gpu_api, torch_gpu):
    """C4: nx=12, nu=6, N=64 (no=384, nc=1536), 1024 instances in one call (the per-GPU batch
    of 8192 is eight such chunks: nothing in a chunk depends on another)."""
    torch = torch_gpu
    from mpcasm.engine import Assembler

    form = problems.random_lti(gpu_api, np.random.default_rng(20262), nx=12, nu=6, N=64)
    B = 1024
    rng = np.random.default_rng(4)
    asm = Assembler(form, batch=B)
    w = rng.uniform(0.1, 1.0, [B, 1, 1])
    asm.set_param("cost", "track s0", "weight", w)            # one weight per instance
    given = torch.as_tensor(rng.normal(0, 0.3, [B, form.given_len]), device="cuda")
    P, q, G, h = asm.assemble(given)
    assert P.shape == (B, 384, 384) and G.shape == (B, 1536, 384)
    assert torch.equal(G, G[:1].expand_as(G))                 # no per-instance arrows
    assert _rel(P.transpose(1, 2), P) <= 1e-13
    # P is affine in the one weight that varies: (P_b - P_0) / (w_b - w_0) is one matrix
    wt = torch.as_tensor(w.ravel(), device="cuda")
    D1 = (P[1] - P[0]) / (wt[1] - wt[0])
    for b in (2, 511, B - 1):
        assert _rel((P[b] - P[0]) / (wt[b] - wt[0]), D1) <= 1e-10
    goal = form.goals["track s0"]
    for b in (0, B - 1):
        goal.update(weight=float(w[b, 0, 0]))
        Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
        assert_close(P[b].cpu().numpy(), Qo, RTOL_TIGHT)
        assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
        assert_close(G[b].cpu().numpy(), Ao, RTOL_TIGHT)
        assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)


def _ltv_batch(api, B, N, rng):
    thetas = rng.uniform(0, 2 * np.pi, B)
    first = [problems.ltv_lipm_steps(api, N=N, theta=float(t)) for t in thetas[:8]]
    # (the symbolic-free closed forms are cheap, but 2048 x 100 of them still take a while:
    # eight true systems, the others scaled copies of them -- every instance still differs)
    A = np.stack([first[i % 8][0] * (1.0 - 1e-3 * (i // 8) / max(B // 8, 1)) for i in range(B)])
    Bm = np.stack([first[i % 8][1] * (1.0 + 1e-3 * (i // 8) / max(B // 8, 1)) for i in range(B)])
    return A, Bm


def test_c5_ltv_fill_at_2048_systems(gpu_api, torch_gpu):
    """C5: per-step (A_k, B_k), N=100, 2048 systems per GPU: the defining recurrence on the
    device, structural zeros, three systems against the oracle's generalisation (parity
    unpinned beyond the LTI case: the reference has no LTV path)."""
    torch = torch_gpu
    from mpcasm import engine

    B, N = 2048, 100
    A, Bm = _ltv_batch(gpu_api, B, N, np.random.default_rng(20263))
    At, Bt = torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda")
    S, U = engine.fill_su(At, Bt, N, ltv=True)
    assert S.shape == (B, N, 3, 3) and U.shape == (B, 1, N, N, 3)
    iu = torch.triu_indices(N, N, offset=1, device="cuda")
    assert U[:, :, iu[0], iu[1], :].abs().max().item() == 0.0          # zeros above the diagonal
    diag = torch.arange(N, device="cuda")
    assert torch.equal(U[:, 0, diag, diag, :], Bt[:, :, :, 0])          # U[k][k] = B_k, exactly
    # U[k][l] = A_k U[k-1][l] (l < k);  S[k]^T = A_k S[k-1]^T
    rec = torch.einsum("bkij,bklj->bkli", At[:, 1:], U[:, 0, :-1])       # (B, N-1, N, 3)
    mask = (torch.arange(N, device="cuda")[None, :] < torch.arange(1, N, device="cuda")[:, None])
    got = U[:, 0, 1:] * mask[None, :, :, None]
    assert _rel(got, rec * mask[None, :, :, None]) <= 1e-13
    St = S.transpose(-1, -2)
    assert _rel(St[:, 1:], torch.matmul(At[:, 1:], St[:, :-1])) <= 1e-13
    assert torch.equal(St[:, 0], At[:, 0])
    for b in (0, 1023, B - 1):
        So, Uo = orc.extend_matrices_ltv(N, A[b], Bm[b])
        assert_close(S[b].cpu().numpy(), So, RTOL_TIGHT)
        assert_close(U[b].cpu().numpy(), np.stack(Uo), RTOL_TIGHT)


def test_ltv_fill_feeds_the_assembly(gpu_api, torch_gpu):
    """LTV end to end: per-instance, per-step (A_k, B_k) -> mpcasm_fill_su(ltv) -> S, U bound as
    per-instance horizon matrices of the biped -> mpcasm_assemble; against the oracle's LTV
    extension + assembly, instance by instance.  (The reference's own time-variant path
    re-extends ONE (A, B) per tick, dynamics.py:222-231: parity unpinned for true LTV.)"""
    torch = torch_gpu
    from mpcasm import engine

    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)
    N, B = conf.horizon_lenght, 96
    rng = np.random.default_rng(77)
    A, Bm = _ltv_batch(gpu_api, B, N, rng)
    S, U = engine.fill_su(torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"), N,
                          ltv=True)
    asm = engine.Assembler(form, batch=B)
    asm.bind_source(("LIP", 0), U[:, 0])
    asm.bind_source(("LIP", 1), S)
    given = rng.normal(0, 0.1, [B, form.given_len])
    P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
    lip = form.dynamics["LIP"]
    saved = list(lip.matrices)
    try:
        for b in (0, 41, B - 1):
            So, Uo = orc.extend_matrices_ltv(N, A[b], Bm[b])
            lip.matrices = Uo + [So]
            lip.update_definitions()
            Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
            assert_close(P[b], Qo, RTOL_TIGHT), assert_close(q[b], qo.ravel(), RTOL_TIGHT)
            assert_close(G[b], Ao, RTOL_TIGHT), assert_close(h[b], ho.ravel(), RTOL_TIGHT)
    finally:
        lip.matrices = saved
        lip.update_definitions()


@pytest.mark.parametrize("jit", [2, 1])
def test_on_chip_horizon_tables_of_an_unstable_badly_conditioned_system(gpu_api, torch_gpu, jit):
    """K1 fused into the assembly does not walk the reference's recurrence step by step (it
    advances four interleaved chains by A^4): checked where a different association of the
    products would show -- spectral radius 1.05, condition number ~1e4, N=24 -- against the
    oracle fed with the reference recurrence's S, U, at the north-star tolerance."""
    torch = torch_gpu
    from mpcasm import capi, engine

    conf = problems.BipedConfig(step_samples=12)           # N = 24, the example as shipped
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([10, 22]), step_count=0)
    N, B = conf.horizon_lenght, 64
    rng = np.random.default_rng(105)
    As, Bs = [], []
    for _ in range(B):
        Q1, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Q2, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        As.append(Q1 @ np.diag([1.05, 0.3, 1.05e-4]) @ Q2)       # cond = 1e4, sigma_max 1.05
        Bs.append(rng.standard_normal((3, 1)))
    A, Bm = np.stack(As), np.stack(Bs)
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        asm = engine.Assembler(form, batch=B, lti=["LIP"])
        asm.bind_lti("LIP", torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda"))
        given = rng.normal(0, 0.1, [B, form.given_len])
        P, q, G, h = (t.cpu().numpy() for t in asm.assemble(given))
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
    lip = form.dynamics["LIP"]
    saved = list(lip.matrices)
    try:
        for b in (0, 31, B - 1):
            So, Uo = orc.extend_matrices(N, A[b], Bm[b])
            lip.matrices = Uo + [So]
            lip.update_definitions()
            Ao, ho, Qo, qo = orc.assemble(form, given[b].reshape(-1, 1))
            assert_close(P[b], Qo, RTOL), assert_close(q[b], qo.ravel(), RTOL)
            assert_close(G[b], Ao, RTOL), assert_close(h[b], ho.ravel(), RTOL)
    finally:
        lip.matrices = saved
        lip.update_definitions()


@pytest.mark.parametrize("jit", [2, 1])
def test_c3_on_the_persistent_kernel(gpu_api, torch_gpu, jit):
    """C3 with its horizon matrices built on chip: one instance no longer fits in LDS with P
    beside the workspace, so the persistent kernel sends the blocks of P straight to HBM (and
    writes the blocks no term reaches as zeros, per instance).  Against the staged pipeline on
    the same inputs (bit for bit where the arithmetic is the same: G, h) and the oracle."""
    torch = torch_gpu
    from mpcasm import capi
    from mpcasm.engine import Assembler

    form = problems.lipm3d(gpu_api, N=32)
    B = 1500
    rng = np.random.default_rng(9)
    given = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        one = Assembler(form, batch=B, lti=["LIP"])
        # garbage in the result buffers first: every element must be written, zeros included
        out = tuple(torch.full_like(t, float("nan")) for t in one.assemble(given))
        P, q, G, h = (t.clone() for t in one.assemble(given, out=out))
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
    assert not any(torch.isnan(t).any().item() for t in (P, q, G, h))
    ref = Assembler(form, batch=B)
    Ps, qs, Gs, hs = ref.assemble(given)
    assert _rel(P, Ps) <= 1e-13 and _rel(q, qs) <= 1e-13 and _rel(G, Gs) <= 1e-13 and _rel(h, hs) <= 1e-13
    for b in (0, B - 1):
        Ao, ho, Qo, qo = orc.assemble(form, given[b].cpu().numpy().reshape(-1, 1))
        assert_close(P[b].cpu().numpy(), Qo, RTOL_TIGHT), assert_close(q[b].cpu().numpy(), qo.ravel(), RTOL_TIGHT)
        assert_close(G[b].cpu().numpy(), Ao, RTOL_TIGHT), assert_close(h[b].cpu().numpy(), ho.ravel(), RTOL_TIGHT)


@pytest.mark.parametrize("jit", [2, 1])
def test_a_large_batch_goes_round_the_workgroups_in_runs_of_four(gpu_api, torch_gpu, jit):
    """From 16 instances per workgroup on, the persistent kernel hands out runs of four consecutive
    instances (whole cache lines of q and h per workgroup) and, once a launch's outputs stream to
    HBM, collects P in LDS: B = 8197 (not a multiple of four, 270 MB of outputs) with a different
    system in every instance, every element pre-set to NaN, against the staged pipeline fed the
    K1 fill's S, U of the same systems."""
    torch = torch_gpu
    from mpcasm import capi, engine

    conf = problems.BipedConfig(step_samples=8)             # N = 16: the C2 shape, no = 36, nc = 76
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)
    N, B = conf.horizon_lenght, 8197
    rng = np.random.default_rng(77)
    get_A, get_B, _ = gpu_api.tools.get_system_matrices("J->CCC")
    taus = rng.uniform(0.09, 0.11, B)
    A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
    Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
    given = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        one = engine.Assembler(form, batch=B, lti=["LIP"])
        one.bind_lti("LIP", A, Bm)
        out = tuple(torch.full_like(t, float("nan")) for t in one.assemble(given))
        P, q, G, h = (t.clone() for t in one.assemble(given, out=out))
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
    assert not any(torch.isnan(t).any().item() for t in (P, q, G, h))
    lib.mpcasm_set_option(capi.OPT_PATH, 2)
    try:
        ref = engine.Assembler(form, batch=B)
        S, U = engine.fill_su(A, Bm, N)
        ref.bind_source(("LIP", 0), U[:, 0])
        ref.bind_source(("LIP", 1), S)
        Ps, qs, Gs, hs = ref.assemble(given)
    finally:
        lib.mpcasm_set_option(capi.OPT_PATH, 0)
    for t, r in ((P, Ps), (q, qs), (G, Gs), (h, hs)):
        per_instance = (t - r).flatten(1).abs().amax(1) / r.flatten(1).abs().amax(1).clamp_min(1e-300)
        assert per_instance.max().item() <= 1e-11, int(per_instance.argmax())


@pytest.mark.parametrize("jit", [2, 1])
def test_csc_data_at_a_batch_that_streams_to_hbm(gpu_api, torch_gpu, jit):
    """The CSC form of the results (plans compiled with csc=) at B = 8197: runs of four instances
    per workgroup, a last run that is not full, a different system in every instance.  Entry for
    entry the dense assembly of the same inputs (bit for bit), every stored entry written."""
    torch = torch_gpu
    from mpcasm import capi, engine

    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)
    B = 8197
    rng = np.random.default_rng(78)
    get_A, get_B, _ = gpu_api.tools.get_system_matrices("J->CCC")
    taus = rng.uniform(0.09, 0.11, B)
    A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
    Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
    given = torch.as_tensor(rng.normal(0, 0.1, [B, form.given_len]), device="cuda")
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        sparse = engine.Assembler(form, batch=B, lti=["LIP"], csc="upper")
        sparse.bind_lti("LIP", A, Bm)
        dense = engine.Assembler(form, batch=B, lti=["LIP"])
        dense.bind_lti("LIP", A, Bm)
        out = tuple(torch.full_like(t, float("nan")) for t in sparse.assemble(given))
        Pd, q, Gd, h = sparse.assemble(given, out=out)
        P, q2, G, h2 = dense.assemble(given)
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)
    c = sparse.csc
    assert Pd.shape == (B, c["pnnz"]) and Gd.shape == (B, c["gnnz"])
    assert not any(torch.isnan(t).any().item() for t in (Pd, q, Gd, h))
    p_flat = torch.as_tensor(c["p_flat"].astype(np.int64), device="cuda")
    g_flat = torch.as_tensor(c["g_flat"].astype(np.int64), device="cuda")
    assert torch.equal(Pd, P.reshape(B, -1)[:, p_flat]) and torch.equal(Gd, G.reshape(B, -1)[:, g_flat])
    assert torch.equal(q, q2) and torch.equal(h, h2)
