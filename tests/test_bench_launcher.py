"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` without torchrun) at world
size 2 on gloo with the kernel calls stubbed (--stub-kernels: no GPU in this container).
What runs for real: the child-process launcher, the rendezvous, mpcasm.dist sharding,
MAX-over-ranks timing, the optional gather and the one-line JSON record."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, "--backend", "gloo", "--stub-kernels", "--steps", "3",
                           "--warmup", "1", "--batch", "8", "--no-cpu-baseline"] + extra,
                          env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def test_gpus_2_starts_two_ranks():
    proc = _run(["--gpus", "2"])
    assert proc.returncode == 0, proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2
    assert rec["config"]["global_batch"] == 16 and rec["config"]["batch_per_gpu"] == 8
    assert rec["gather"]["instances"] == 16                 # both shards arrived, in rank order
    assert rec["value"] > 0 and rec["steps"] == 3


def test_single_rank_needs_no_launcher():
    proc = _run(["--gpus", "1"])
    assert proc.returncode == 0, proc.stderr
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and "gather" not in rec


def test_a_failed_rank_fails_the_run():
    proc = _run(["--gpus", "2", "--stub-fail-rank", "1"], timeout=120)
    assert proc.returncode != 0
    assert "rank 1 exited" in proc.stderr


def test_world_size_mismatch_is_an_error():
    proc = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert proc.returncode != 0 and "WORLD_SIZE" in proc.stderr


def test_ranks_seen_are_in_the_record():
    proc = _run(["--gpus", "2"])
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_ranks_seen"] == 2


def test_one_rank_through_the_process_group():
    """--dist-at-world-1: a single rank still goes through init_process_group, the barriers, the
    MAX reduction and the gather (on the GPU box the same flag runs them over RCCL)."""
    proc = _run(["--gpus", "1", "--dist-at-world-1"])
    assert proc.returncode == 0, proc.stderr
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1
    assert rec["gather"]["instances"] == 8


def test_under_a_launcher_the_world_comes_from_the_environment():
    """`torchrun --nproc_per_node 2 bench.py` without --gpus: the ranks the launcher started."""
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        e = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen(
            [sys.executable, BENCH, "--backend", "gloo", "--stub-kernels", "--steps", "2", "--warmup", "1",
             "--batch", "4", "--no-cpu-baseline"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
            text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    rec = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]     # rank 0 alone reports


def test_devices_are_counted_without_the_runtime(monkeypatch):
    """The launcher's parent counts GPUs from the environment / the driver's topology: importing
    torch is not needed (and HIP is never initialised in the process that starts the ranks)."""
    sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
    from mpcasm import dist as mdist

    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert mdist.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert mdist.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    for var in ("ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert mdist.visible_gpus() in (None, 0) or mdist.visible_gpus() > 0
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("# workload")]
    assert "import torch" not in body and "device_count" not in body
