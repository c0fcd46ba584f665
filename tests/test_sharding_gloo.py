"""CPU, 2 processes (gloo): the multi-GPU layer -- batch sharding without a data-path
collective and the optional gather of the assembled QPs.  The per-rank "assembly"
is the oracle here (no GPU in this container); on the GPU box the same layer runs
over RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpcasm.dist import gather_batch, instance_seed, local_shard, max_over_ranks, shard_bounds


def test_shard_bounds_cover_the_batch():
    for batch in (0, 1, 7, 8, 4096, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert instance_seed(5, 3) == instance_seed(5, 3) != instance_seed(5, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "mpc-interface_amd"), root, here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpcasm import problems
        from oracle import qp_oracle as orc
        import mpc_interface.tools as tools

        tools.extend_matrices = orc.extend_matrices        # host logic only (CPU test)
        api = problems.load_api("mpc_interface")
        form = problems.body_case(api)
        # every instance's inputs depend on its global index only
        given = np.stack([np.random.default_rng(instance_seed(11, i)).standard_normal(form.given_len)
                          for i in range(batch)])
        mine = local_shard(given, world, rank)
        lo, hi = shard_bounds(batch, world, rank)
        assert mine.shape[0] == hi - lo
        P = np.stack([orc.assemble(form, g.reshape(-1, 1))[2] for g in mine]) if len(mine) else \
            np.zeros((0, form.optim_len, form.optim_len))
        q = np.stack([orc.assemble(form, g.reshape(-1, 1))[3].ravel() for g in mine]) if len(mine) \
            else np.zeros((0, form.optim_len))
        allP = gather_batch(torch.as_tensor(P), batch)
        allq = gather_batch(torch.as_tensor(q), batch)
        slowest = max_over_ranks(1.0 + rank)
        assert slowest == float(world)
        np.save(os.path.join(out_dir, "P%d.npy" % rank), allP.numpy())
        np.save(os.path.join(out_dir, "q%d.npy" % rank), allq.numpy())
        np.save(os.path.join(out_dir, "given.npy"), given)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [6, 5])
def test_two_ranks_assemble_and_gather(tmp_path, batch, cpu_api):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    from mpcasm import problems
    from oracle import qp_oracle as orc

    form = problems.body_case(cpu_api)
    given = np.load(tmp_path / "given.npy")
    want_q = np.stack([orc.assemble(form, g.reshape(-1, 1))[3].ravel() for g in given])
    for rank in range(world):
        P, q = np.load(tmp_path / ("P%d.npy" % rank)), np.load(tmp_path / ("q%d.npy" % rank))
        assert P.shape == (batch, form.optim_len, form.optim_len)
        assert np.array_equal(q, want_q)                    # batch order restored on every rank
        assert np.array_equal(P[0], P[-1])                  # P does not depend on `given` here
