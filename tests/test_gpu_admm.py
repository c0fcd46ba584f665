"""mpcasm_admm (K5, SURVEY.md section 8 f3's "or"): OSQP's ADMM iteration on the dense QPs where
mpcasm_assemble leaves them, against oracle/admm_oracle.py (the numpy restatement of the published
iteration) iterate by iterate, and against the KKT conditions of the QPs at the fixed point."""
import numpy as np
import pytest

from helpers import assert_close
from mpcasm import problems
from oracle import admm_oracle as ao

pytestmark = pytest.mark.gpu
TOL = 1e-10      # the north star's tolerance; observed ~1e-13 (an explicit inverse against numpy's LU solve)


@pytest.fixture
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def random_qps(rng, B, no, nc):
    R = rng.standard_normal((B, no + 3, no))
    P = np.einsum("bki,bkj->bij", R, R) + 0.1 * np.eye(no)
    q = rng.standard_normal((B, no))
    G = rng.standard_normal((B, nc, no))
    h = rng.uniform(0.1, 1.0, (B, nc))
    return P, q, G, h


def to_dev(torch, *arrays):
    return [torch.as_tensor(np.ascontiguousarray(a), device="cuda") for a in arrays]


@pytest.mark.parametrize("no,nc", [(36, 76), (5, 3), (70, 10), (64, 65), (33, 200), (7, 0)],
                         ids=["biped", "tiny", "wide", "edge", "tall", "free"])
def test_iterates_against_the_oracle(gpu_api, torch_gpu, no, nc):
    """0, 1, 7 and 40 iterations from a cold start, 25 more from a warm one: x, y, z and the two residuals
    of every instance equal the oracle's (more unknowns than a wavefront has lanes, more limits than
    unknowns and fewer, no limits at all)."""
    torch = torch_gpu
    from mpcasm import engine

    rng = np.random.default_rng(no * 1000 + nc)
    B = 9
    P, q, G, h = random_qps(rng, B, no, nc)
    dP, dq, dG, dh = to_dev(torch, P, q, G, h)
    for iters in (0, 1, 7, 40):
        x, y, z, res = engine.admm(dP, dq, dG, dh, iters=iters, rho=1.0)
        for b in range(B):
            xo, yo, zo, ro = ao.admm(P[b], q[b], G[b], h[b], iters=iters, rho=1.0)
            assert_close(x[b].cpu().numpy(), xo, TOL, "x"), assert_close(y[b].cpu().numpy(), yo, TOL, "y")
            assert_close(z[b].cpu().numpy(), zo, TOL, "z")
            assert np.allclose(res[b].cpu().numpy(), ro, rtol=1e-6, atol=1e-12)
    # warm: 25 more in place, OSQP's default steps
    x0, y0, z0, _ = engine.admm(dP, dq, dG, dh, iters=15)
    keep = [t.clone() for t in (x0, y0, z0)]
    x1, y1, z1, _ = engine.admm(dP, dq, dG, dh, x0, y0, z0, iters=25)
    assert x1.data_ptr() == x0.data_ptr()
    for b in (0, B - 1):
        start = [t[b].cpu().numpy() for t in keep]
        xo, yo, zo, _ = ao.admm(P[b], q[b], G[b], h[b], *start, iters=25)
        assert_close(x1[b].cpu().numpy(), xo, TOL, "x warm"), assert_close(y1[b].cpu().numpy(), yo, TOL, "y warm")
        assert_close(z1[b].cpu().numpy(), zo, TOL, "z warm")


def test_from_the_assembly_to_the_solution_for_a_fleet_of_bipeds(gpu_api, torch_gpu):
    """The walking loop's tick for 4 096 walkers without leaving the device: mpcasm_assemble, then
    mpcasm_admm on its buffers (biped_mpc_loop.py:50-60).  64 sampled walkers against the oracle's
    iteration on the oracle's own matrices; every walker's fixed point satisfies the KKT conditions."""
    torch = torch_gpu
    from mpcasm import engine
    from oracle import qp_oracle as orc

    form = problems.biped(gpu_api, problems.BipedConfig(step_samples=8))
    form.update(step_times=np.array([6, 14]), step_count=0)
    B = 4096
    rng = np.random.default_rng(8)
    given = rng.normal(0, 0.001, [B, form.given_len])
    asm = engine.Assembler(form, batch=B)
    P, q, G, h = asm.assemble(given)
    x, y, z, res = engine.admm(P, q, G, h, iters=60, rho=1.0)
    for b in rng.choice(B, 64, replace=False):
        Go, ho, Po, qo = orc.assemble(form, given[b].reshape(-1, 1))
        xo, yo, zo, ro = ao.admm(Po, qo, Go, ho, iters=60, rho=1.0)
        assert_close(x[b].cpu().numpy(), xo, TOL, "x"), assert_close(y[b].cpu().numpy(), yo, TOL, "y")
    # on to the fixed point, warm
    for _ in range(3):
        x, y, z, res = engine.admm(P, q, G, h, x, y, z, iters=500, rho=1.0)
    assert float(res.max()) < 1e-12
    Gx = torch.einsum("brc,bc->br", G, x)
    assert float((Gx - h).max()) < 1e-11 and float(y.min()) > -1e-11
    assert float((y * (Gx - h)).abs().max()) < 1e-11
    grad = torch.einsum("bij,bj->bi", P, x) + q + torch.einsum("brc,br->bc", G, y)
    assert float(grad.abs().max()) < 1e-11
    assert int((y > 1e-9).sum(dim=1).min()) >= 1          # (limits bind for every walker)


def test_the_inverse_kept_from_one_call_to_the_next(gpu_api, torch_gpu):
    """The same model with a new `given`: P and G as before, q and h new (body.py:236-302).  A call that is handed
    the K^-1 an earlier call wrote gives the iterates of a call that factors again, bit for bit -- and does not
    look at P any more."""
    torch = torch_gpu
    from mpcasm import engine

    rng = np.random.default_rng(5)
    B, no, nc = 33, 36, 76
    P, q, G, h = to_dev(torch, *random_qps(rng, B, no, nc))
    kinv = torch.full((B, no, no), float("nan"), dtype=torch.float64, device="cuda")
    x0, y0, z0, _ = engine.admm(P, q, G, h, iters=20, rho=1.0, kinv=kinv)
    assert not bool(torch.isnan(kinv).any())
    Kref = np.linalg.inv(P[3].cpu().numpy() + 1e-6 * np.eye(no) + G[3].cpu().numpy().T @ G[3].cpu().numpy())
    assert_close(kinv[3].cpu().numpy(), Kref, 1e-9, "K^-1")
    q2, h2 = to_dev(torch, rng.standard_normal((B, no)), rng.uniform(0.1, 1.0, (B, nc)))
    fresh = engine.admm(P, q2, G, h2, iters=30, rho=1.0)
    garbage = torch.full_like(P, float("nan"))                 # (not read with a valid inverse and no residuals)
    kept = engine.admm(garbage, q2, G, h2, iters=30, rho=1.0, kinv=kinv, kinv_valid=True, residuals=False)
    for a, b in zip(fresh[:3], kept[:3]):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        engine.admm(P, q2, G, h2, kinv_valid=True)
    with pytest.raises(ValueError):
        engine.admm(P, q2, G, h2, kinv=kinv[:, :5].contiguous())


def test_what_the_kernel_refuses(gpu_api, torch_gpu):
    torch = torch_gpu
    from mpcasm import capi, engine

    rng = np.random.default_rng(2)
    P, q, G, h = to_dev(torch, *random_qps(rng, 3, 6, 4))
    with pytest.raises(ValueError):
        engine.admm(P, q, G, h, x=torch.zeros((3, 6), dtype=torch.float64, device="cuda"))   # a warm start takes all three
    with pytest.raises(ValueError):
        engine.admm(P, q[:, :5].contiguous(), G, h)
    for kw in (dict(rho=0.0), dict(sigma=-1.0), dict(alpha=2.0), dict(iters=-1)):
        with pytest.raises(capi.MpcasmError) as err:
            engine.admm(P, q, G, h, **kw)
        assert err.value.status == -1
    # C3's QP (96 unknowns, 196 limits) does not fit in LDS: said so, nothing launched
    big = to_dev(torch, *random_qps(rng, 2, 96, 196))
    with pytest.raises(capi.MpcasmError) as err:
        engine.admm(*big)
    assert err.value.status == capi.ERR_LIMIT
    # an indefinite P + sigma I + rho G'G: NaNs for that instance, the others untouched by it
    P2 = P.clone()
    P2[1] = -50.0 * torch.eye(6, dtype=torch.float64, device="cuda")
    x, y, z, res = engine.admm(P2, q, G, h, iters=5)
    assert bool(torch.isnan(x[1]).all()) and bool(torch.isnan(res[1]).all())
    assert not bool(torch.isnan(x[0]).any()) and not bool(torch.isnan(x[2]).any())
