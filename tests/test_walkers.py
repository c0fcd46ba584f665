"""The batched tick driver: vectorised clock, step indicator and
stepping centres against the single-walker host logic (CPU), and a fleet of walkers in
different phases against the oracle stepped walker by walker (GPU)."""
import numpy as np
import pytest

from helpers import RTOL_TIGHT, assert_close, golden
from mpcasm import problems
from mpcasm.walkers import FleetClock, WalkerFleet, step_indicator, stepping_centers, \
    steps_in_preview
from oracle import qp_oracle as orc
import mpc_interface.tools as tools


def test_fleet_clock_matches_the_single_walker_clock():
    n, phases = 8, np.arange(11) % 8
    fleet = FleetClock(n, 2, phases)
    singles = []
    for phi in phases:
        c = problems.StepClock(n, 2)
        for _ in range(phi):
            c.tick()
        singles.append(c)
    for _ in range(20):
        for b, c in enumerate(singles):
            assert np.array_equal(fleet.step_times[b], c.step_times)
            assert fleet.step_count[b] == c.step_count
        fleet.tick()
        for c in singles:
            c.tick()


def test_step_indicator_and_centres_match_tools():
    N = 16
    for times in (np.array([6, 14]), np.array([7, 15]), np.array([0, 8]), np.array([3, 11])):
        keep = steps_in_preview(times, N)
        E = step_indicator(times[keep][None, :], N)[0]
        assert np.array_equal(E, tools.plan_steps(N, 0, step_times=times))
    for count in range(4):
        for p in (1, 2, 3):
            c = stepping_centers(np.array([count]), p, [0.0, 0.28])[0]
            assert_close(c, tools.find_step_centers(count, p, [0.0, 0.28]), 0)


def test_bucket_widths_follow_the_golden_tick_sequence():
    """One walker in phase 0 reproduces the reference's shape sequence (34/36 wide)."""
    g = golden("g3_biped_N16")
    clock = FleetClock(8, 2, [0])
    for tick in range(18):
        p = int(steps_in_preview(clock.step_times, 16).sum())
        assert 32 + 2 * p == g["shapes"][tick][2]
        clock.tick()


@pytest.mark.gpu
def test_fleet_against_the_oracle(gpu_api):
    conf = problems.BipedConfig(step_samples=8)
    batch = 19
    phases = np.arange(batch) % 8
    fleet = WalkerFleet(batch, phases=phases, conf=conf, api=gpu_api)
    ref = problems.biped(gpu_api, conf)          # one formulation, re-pointed at each walker
    clocks = []
    for phi in phases:
        c = problems.StepClock(conf.step_samples, 2)
        for _ in range(phi):
            c.tick()
        clocks.append(c)
    rng = np.random.default_rng(3)
    widths = set()
    for tick in range(10):
        given = rng.normal(0, 0.1, [batch, fleet.given_len])
        results = fleet.tick(given)
        seen = np.zeros(batch, dtype=bool)
        for res in results:
            P, q, G, h = (res[k].cpu().numpy() for k in ("P", "q", "G", "h"))
            widths.add(P.shape[1])
            for row, b in enumerate(res["index"]):
                seen[b] = True
                if tick in (0, 3, 9) or b < 3:
                    ref.update(step_times=clocks[b].step_times, step_count=clocks[b].step_count)
                    A, hh, Q, qq = orc.assemble(ref, given[b].reshape(-1, 1))
                    assert Q.shape[0] == P.shape[1] == 32 + 2 * res["p"]
                    assert_close(P[row], Q, RTOL_TIGHT, "P")
                    assert_close(q[row], qq.ravel(), RTOL_TIGHT, "q")
                    assert_close(G[row], A, RTOL_TIGHT, "G")
                    assert_close(h[row], hh.ravel(), RTOL_TIGHT, "h")
        assert seen.all()
        for c in clocks:
            c.tick()
    assert widths == {34, 36}


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [0, 2], ids=["per-plan kernels", "ahead-of-time kernel"])
def test_fleet_ticks_replayed_from_graphs(gpu_api, jit):
    """WalkerFleet(graphs=True): a tick is a copy of `given` and one hipGraph launch; over two
    step cycles (capture in the first, replay in the second) every walker's QP is the one the
    fleet computes launch by launch.  On the ahead-of-time kernel the 34- and 36-unknown buckets
    share one instantiation with different LDS sizes: its residency is kept per size and its
    dynamic-LDS limit only ever raised, so capture and replay make no runtime call between them."""
    import torch

    from mpcasm import capi

    conf = problems.BipedConfig(step_samples=8)
    batch = 600
    phases = np.arange(batch) % 8
    lib = capi.load()
    lib.mpcasm_set_option(capi.OPT_JIT, jit)
    try:
        plain = WalkerFleet(batch, phases=phases, conf=conf, api=gpu_api)
        replayed = WalkerFleet(batch, phases=phases, conf=conf, api=gpu_api, graphs=True)
        # ... the buckets as parallel branches of the graph, each with its share of the workgroup slots, and
        # `given` written straight into the fleet's own buffer (no copy in front of the replay)
        branches = WalkerFleet(batch, phases=phases, conf=conf, api=gpu_api, graphs=True, side_by_side=True)
        own = branches.given_buffer()
        rng = np.random.default_rng(4)
        for tick in range(2 * 2 * conf.step_samples + 3):
            given = torch.as_tensor(rng.normal(0, 0.1, [batch, plain.given_len]), device="cuda")
            own.copy_(given)
            a, b, c = plain.tick(given), replayed.tick(given), branches.tick(own)
            assert [r["p"] for r in a] == [r["p"] for r in b] == [r["p"] for r in c]
            for ra, rb, rc in zip(a, b, c):
                assert np.array_equal(ra["index"], rb["index"]) and np.array_equal(ra["index"], rc["index"])
                for k in ("P", "q", "G", "h"):
                    assert torch.equal(ra[k], rb[k]), (tick, k)
                    assert torch.equal(ra[k], rc[k]), (tick, k, "branches")
    finally:
        lib.mpcasm_set_option(capi.OPT_JIT, 0)


@pytest.mark.gpu
def test_a_fleet_of_4096_walkers_against_the_oracle(gpu_api):
    """4096 walkers (buckets of 512 and 3584: the kernels compiled per plan), a full step cycle and a
    half: every tick 64 walkers -- the first and last of every bucket and random ones -- against the
    oracle stepped walker by walker.  The buckets take their walkers' rows of `given` by index inside
    the kernel (mpcasm_assemble_indexed) and the parameters of the place in the cycle as they are."""
    import torch

    conf = problems.BipedConfig(step_samples=8)
    batch = 4096
    phases = np.arange(batch) % 8
    fleet = WalkerFleet(batch, phases=phases, conf=conf, api=gpu_api)
    ref = problems.biped(gpu_api, conf)
    clock = FleetClock(conf.step_samples, conf.num_steps, phases)
    rng = np.random.default_rng(5)
    for tick in range(12):
        given = rng.normal(0, 0.1, [batch, fleet.given_len])
        results = fleet.tick(torch.as_tensor(given, device="cuda"))
        assert sorted(int(b) for res in results for b in res["index"]) == list(range(batch))
        for res in results:
            assert "persistent" in fleet.buckets[res["p"]]["asm"].last_kernel()
            idx = res["index"]
            rows = sorted({0, idx.size - 1} | set(int(x) for x in rng.integers(0, idx.size, 32)))
            P, q, G, h = (res[k][rows].cpu().numpy() for k in ("P", "q", "G", "h"))
            for at, row in enumerate(rows):
                b = int(idx[row])
                ref.update(step_times=clock.step_times[b], step_count=int(clock.step_count[b]))
                A, hh, Q, qq = orc.assemble(ref, given[b].reshape(-1, 1))
                assert Q.shape[0] == P.shape[1] == 32 + 2 * res["p"]
                assert_close(P[at], Q, RTOL_TIGHT, "P"), assert_close(q[at], qq.ravel(), RTOL_TIGHT, "q")
                assert_close(G[at], A, RTOL_TIGHT, "G"), assert_close(h[at], hh.ravel(), RTOL_TIGHT, "h")
        clock.tick()


@pytest.mark.gpu
def test_rows_of_given_by_index(gpu_api):
    """Assembler.assemble(index=...): instance b reads row index[b] of a larger `given` -- the same
    numbers as gathering first; on a plan that is not on the persistent kernel the engine gathers."""
    import torch

    from mpcasm import capi, engine

    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(gpu_api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)
    rng = np.random.default_rng(6)
    B, fleet = 700, 5000
    given = torch.as_tensor(rng.normal(0, 0.1, [fleet, form.given_len]), device="cuda")
    index = torch.as_tensor(rng.permutation(fleet)[:B].astype(np.int32), device="cuda")
    for path in (0, 2):
        asm = engine.Assembler(form, batch=B)
        asm.set_option(capi.OPT_PATH, path)
        want = tuple(t.clone() for t in asm.assemble(given.index_select(0, index.long())))
        got = asm.assemble(given, index=index)
        assert ("persistent" in asm.last_kernel()) == (path == 0)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        part = asm.assemble(given, index=index, count=B - 5)
        assert torch.equal(part[0][:B - 5], want[0][:B - 5])
    with pytest.raises(ValueError):
        asm.assemble(given, index=index.long())
