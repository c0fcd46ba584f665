"""Multi-GPU layer: one process per GPU, batch sharded with no collective on the
data path (QP instances are independent).

The only communication is optional and happens after the assembly: gathering the
assembled QPs of every rank (``gather_batch``, an all-gather over RCCL/xGMI when
the process group's backend is ``nccl``; ``gloo`` on CPU in the tests) and the
reduction of timings in ``bench.py``.  On an MI355X node every GPU pair has its
own xGMI link, so one all-gather of ``B/world`` instances per rank moves
``(world-1)/world`` of the batch into each GPU over seven links in parallel.
"""
import numpy as np


def shard_bounds(batch, world_size, rank):
    """Contiguous slice ``[lo, hi)`` of ``batch`` instances owned by ``rank``; the
    first ``batch % world_size`` ranks take one instance more."""
    base, extra = divmod(int(batch), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def local_shard(array, world_size, rank):
    """This rank's rows of a per-instance array (numpy or torch)."""
    lo, hi = shard_bounds(array.shape[0], world_size, rank)
    return array[lo:hi]


def gather_batch(local, batch, group=None):
    """All-gather per-instance tensors sharded by :func:`shard_bounds` back into
    batch order on every rank.  ``local``: tensor ``(hi - lo, ...)``; returns
    ``(batch, ...)``.

    Equal shards (the usual case: the batch is a multiple of the world size) go straight into
    the result with ONE ``all_gather_into_tensor`` -- every rank's slice lands at its final
    place, no staging copy, no concatenation.  Ragged shards are padded to the largest one
    for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [b - a for a, b in (shard_bounds(batch, world, r) for r in range(world))]
    longest = max(sizes)
    on_host = local.is_cuda and dist.get_backend(group) == "gloo"
    if min(sizes) == longest and not on_host:
        local = local.contiguous()
        out = torch.empty((batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]),
                          dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad])
    padded = padded.contiguous()
    if on_host:
        # gloo has no all-gather of device tensors: stage through the host (rehearsals of the
        # multi-rank path on a box whose ranks share one GPU; RCCL takes the branches above / below)
        host = padded.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return torch.cat([p[:n] for p, n in zip(parts, sizes)]).to(local.device)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)])


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT touching the HIP runtime (a parent that
    starts one rank per GPU must not initialise the device it hands out): the compute nodes of
    the kernel driver's topology, narrowed by ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES``
    when set.  ``None`` when the topology cannot be read (let the ranks find out)."""
    import glob
    import os

    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        value = os.environ.get(var)
        if value is not None:
            return len([x for x in value.split(",") if x.strip() != ""])
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:      # (CPU nodes have none)
            count += 1
    return count


def max_over_ranks(value, device=None, group=None):
    """MAX-reduce a python float over the ranks (timing of the bench)."""
    import torch
    import torch.distributed as dist

    if device is not None and dist.get_backend(group) == "gloo":
        device = None           # (host tensor: every gloo build reduces those)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def instance_seed(base_seed, index):
    """Seed of one instance's synthetic inputs: independent of how the batch is
    sharded, so every world size assembles the same QPs."""
    return int(np.random.SeedSequence([int(base_seed), int(index)]).generate_state(1)[0])
