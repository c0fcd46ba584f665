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
