"""GPU: the multi-rank plumbing over RCCL with the ONE GPU at hand: `bench.py --dist-at-world-1`
initialises the nccl process group with a single rank and sends the barriers, the MAX
reduction, the rank roll-call and the gather of the assembled QPs through it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_with_one_rank():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--dist-at-world-1", "--backend", "nccl",
         "--steps", "50", "--warmup", "5", "--batch", "1024", "--no-cpu-baseline"],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1 and len(rec["devices"]) == 1
    assert rec["devices"][0].startswith("r0 cuda:0")
    g = rec["gather"]
    assert g["backend"] == "nccl" and g["instances_per_gpu_after"] == 1024 and g["ms"] > 0
    assert rec["value"] > 0 and "resident" in rec["roofline"]["kernel"]
    # the ONE line of the contract survives a 2000-character tail, sub-records included
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0]
    assert len(line) < 2000, len(line)
    for key in ("roofline", "fill", "variants", "c3", "c4", "c5", "admm", "f2", "B65536"):
        assert key in rec, key
    assert "error" not in rec["c4"] and rec["c4"]["shared_ms"] < rec["c4"]["general_ms"]
    assert "error" not in rec["c5"] and "error" not in rec["admm"] and rec["admm"]["iter_us"] > 0
    assert rec["kernarg_dev"] == "1"          # (kernel arguments in device memory: set by bench.py itself)
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1


def test_gather_batch_over_rccl_in_process():
    """mpcasm.dist.gather_batch on device tensors through the nccl backend (world 1): the
    all_gather_into_tensor path writes the result in place."""
    import torch
    import torch.distributed as dist

    from mpcasm import dist as mdist

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x = torch.arange(24, dtype=torch.float64, device="cuda").reshape(6, 4)
        out = mdist.gather_batch(x, 6)
        assert out.is_cuda and torch.equal(out, x) and out.data_ptr() != x.data_ptr()
        assert mdist.max_over_ranks(3.5, device=x.device) == 3.5
    finally:
        dist.destroy_process_group()
