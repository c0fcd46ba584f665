#!/usr/bin/env python3
"""Benchmark of the QP-assembly hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): the biped-LIPM walking
formulation, N=16, 2 axes, 6 costs, 3 boxes (36 unknowns, 76 inequality rows in
the headline phase), B independent instances per GPU in fp64, synthetic inputs
resident in HBM.  One *step* is one pass of the hot path over one batch:

    K1  horizon matrices S, U from the per-instance (A, B)          (tools.py:14-33)
    K2-K4  S, U, given, parameters -> P, q, G, h for every instance (body.py:142-348)

by default ONE launch of mpcasm_assemble that builds the horizon matrices on chip
(--two-kernels: mpcasm_fill_su + mpcasm_assemble through S, U in HBM).  The outputs
rotate over --rotate buffer sets so that at B=4096 (140 MB per step) consecutive steps
do not rewrite lines that still sit in the 256 MiB Infinity Cache.

Prints ONE JSON line (rank 0): whole-job assemblies/s, the roofline object of the
dominant kernel (one pair of hipEvents on the launch stream around the timed region: the
same clock as ms_per_step), `fill` sub-records (K1 alone, timed after the main region, on
the north-star shapes), an `extra` record at B=65536 (a working set far beyond the
Infinity Cache), for N>1 a separately timed `gather` record (all-gather of the assembled
QPs over RCCL, never part of `value`) and, at N=1, the CPU baseline (the numpy oracle --
a port of the reference algorithm -- on this box's host cores, bounded sample).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); the batch is
sharded with no collective on the data path (weak scaling: B per GPU is fixed); the only
collectives inside the measurement are the barriers around the timed region and the MAX
of the elapsed time.  `python bench.py --gpus N` without a launcher starts the N ranks
itself (child processes, before this process has touched the GPU); under
`python -m torch.distributed.run` the ranks come from the environment.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the CPU baseline is quoted for one core per process: keep BLAS single-threaded
# kernel arguments in device memory (a documented setting of the HIP runtime, read when it initialises):
# by default they live in host memory, and the first loads of EVERY launch -- the kernel's own arguments --
# cross PCIe: 2.5 us of a 27 us launch ("set-up")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (os.path.join(ROOT, "mpc-interface_amd"), ROOT):
    if path not in sys.path:
        sys.path.insert(0, path)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
METRIC = "QP assemblies/sec (P,q,G,h), biped N=16 batched"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="untimed steps before the warm-up until this much time has passed: the "
                         "device's clocks take 15-25 ms of load to settle (tools/launch_series.py: "
                         "the first launches after idle run up to 15 %% slower)")
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--rotate", type=int, default=4, help="output buffer sets the steps cycle over")
    ap.add_argument("--streams", type=int, default=2,
                    help="launch streams the (independent) steps are dealt to in turn: a launch starts on "
                         "the CUs the one before has already left instead of waiting for its last "
                         "workgroup and a launch gap (tools/two_streams.py: 30.4 -> 26.6 us per step); 1 = "
                         "one stream.  The roofline record is always taken on ONE stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the fill / B=65536 / gather sub-records (main line only)")
    ap.add_argument("--two-kernels", action="store_true",
                    help="step = mpcasm_fill_su (S, U through HBM) + mpcasm_assemble instead of the "
                         "default single launch that builds the horizon matrices on chip")
    ap.add_argument("--fused", action="store_true", help="(the default; kept for scripts)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dist-at-world-1", action="store_true",
                    help="with ONE rank, still initialise the process group (nccl = RCCL) and run the "
                         "barriers, the MAX reduction and the gather of the assembled QPs through it: "
                         "RCCL start-up, stream semantics and the record plumbing on the hardware at hand")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="TEST ONLY: this rank exits 3")
    ap.add_argument("--stub-kernels", action="store_true",
                    help="TEST ONLY (tests/test_bench_launcher.py): no GPU, no kernels -- exercises "
                         "the rank launcher, sharding, gather and record plumbing on CPU tensors")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------
# rank launcher
# --------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rendezvous_port(world):
    """A port of its own only for a world of ONE rank (--dist-at-world-1): with several ranks every
    rank must be told the same one by the launcher -- a default picked per rank would make
    init_process_group hang until it times out instead of failing at once."""
    if "MASTER_PORT" in os.environ:
        return
    if world > 1:
        raise SystemExit("bench.py: %d ranks but no MASTER_PORT in the environment (start the ranks with "
                         "torch.distributed.run, or let `bench.py --gpus N` start them)" % world)
    os.environ["MASTER_PORT"] = str(_free_port())


def launch_ranks(args, argv):
    """Start ``args.gpus`` ranks of this script as child processes (one per GPU) and wait.
    The parent never initialises the GPU -- it counts devices from the kernel driver's topology,
    not through torch or HIP -- and rank 0's stdout (the JSON line) passes through.  Returns the
    exit code: non-zero if any rank failed (the others are stopped); the ranks are stopped too
    when the parent is interrupted or terminated."""
    import signal

    n = args.gpus
    if not args.stub_kernels and args.backend == "nccl":
        from mpcasm.dist import visible_gpus

        have = visible_gpus()
        if have is not None and have < n:
            print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []

    def stop_all(*_):
        for proc in procs:
            if proc.poll() is None:
                proc.terminate()

    previous = signal.signal(signal.SIGTERM, lambda *a: (stop_all(), sys.exit(143)))
    rc = 0
    try:
        for rank in range(n):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=None if rank == 0 else subprocess.DEVNULL))
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with %d; stopping the others" % (r, code),
                          file=sys.stderr)
                    for o in pending:
                        procs[o].terminate()
            time.sleep(0.05)
    finally:
        stop_all()
        signal.signal(signal.SIGTERM, previous)
    return rc


# --------------------------------------------------------------------------
# workload
# --------------------------------------------------------------------------
def build_workload(batch, seed, step_times=(6, 14), reduced=False):
    """The biped formulation in its 36-wide phase + per-instance synthetic inputs.
    ``step_times`` (7, 15): the phase with ONE previewed step -- the 34-wide bucket;
    ``reduced``: without the zero-weight terminal cost and the terminal box."""
    from mpcasm import engine, problems

    api = problems.load_api("mpc_interface")
    conf = problems.BipedConfig(step_samples=8)               # N = 16
    form = problems.biped(api, conf, reduced=reduced)
    form.update(step_times=np.array(step_times), step_count=0)   # phase phi=1: no=36, nc=76
    N = conf.horizon_lenght

    rng = np.random.default_rng(seed)
    given = np.zeros([batch, form.given_len])
    for var, ids in form.given_ID.items():
        if var.startswith("x0"):
            given[:, ids] = rng.normal(0, 0.05, [batch, len(ids)])
        elif var.startswith("s0"):
            given[:, ids] = rng.uniform(-0.1, 0.1, [batch, len(ids)])
        else:
            given[:, ids] = rng.normal(0, 0.01, [batch, len(ids)])
    # per-instance dynamics: jerk-input LIPM sampled at a per-instance period
    get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
    taus = rng.uniform(0.09, 0.11, batch)
    A = np.stack([get_A(tau=t) for t in taus])
    B = np.stack([get_B(tau=t) for t in taus])
    aims = rng.uniform(0, 0.6, [batch, 1, 1])
    return dict(api=api, conf=conf, form=form, N=N, given=given, A=A, B=B, aims=aims,
                engine=engine, problems=problems)


def cpu_baseline(work, budget_s=12.0):
    """The oracle (numpy port of the reference path) on one host core: per
    instance extend_matrices + preview matrices + all QP blocks."""
    from oracle import qp_oracle as orc

    form, N = work["form"], work["N"]
    lip = form.dynamics["LIP"]
    saved = list(lip.matrices)
    done, t0 = 0, time.perf_counter()
    while True:
        b = done % work["given"].shape[0]
        S, U = orc.extend_matrices(N, work["A"][b], work["B"][b])
        lip.matrices = U + [S]
        lip.update_definitions()
        form.goals["track vel_x"].update(aim=work["aims"][b, 0])
        maps = orc.qp_index_maps(form.domain, form.optim_variables)
        PM = orc.preview_matrices(form, maps)
        orc.assemble(form, work["given"][b].reshape(-1, 1), PM, maps)
        done += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s:
            break
    lip.matrices = saved
    lip.update_definitions()
    return done / elapsed, done, elapsed


def cpu_baseline_compiled(work, threads, budget_s=4.0):
    """The same per-instance path compiled: oracle/assemble_port.c (plain C restatement of
    extend_matrices + make_preview_matrices + generate_all_qp_matrices, body.py:142-348, the dense
    matrices and loop order of the numpy oracle) on ``threads`` host threads, every thread on its own
    slice of the instances (the library call releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import c_port

    form = work["form"]
    recipe = c_port.Recipe(form, per_instance="LIP")
    count = work["given"].shape[0]
    consts = np.tile(recipe.consts, (count, 1))
    # (the workload's per-instance aim of the velocity cost; a plain cost's cross_aim IS its aim, goal.py:67-86)
    for field in ("aim", "cross_aim"):
        consts[:, recipe.const_slice("cost", "track vel_x", field)] = np.asarray(work["aims"]).reshape(count, -1)[:, 0:1]
    A, B, given = work["A"], work["B"], work["given"]
    recipe.assemble(A[:8], B[:8], given[:8], consts[:8], keep=False)

    def loop(t):
        lo, hi = t * count // threads, (t + 1) * count // threads
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            recipe.assemble(A[lo:hi], B[lo:hi], given[lo:hi], consts[lo:hi], keep=False)
            done += hi - lo
        return done, time.perf_counter() - t0

    with ThreadPoolExecutor(threads) as pool:
        results = list(pool.map(loop, range(threads)))
    return sum(d / t for d, t in results), sum(d for d, _ in results)


def cpu_fill_compiled(budget_s=2.0):
    """K1 on one host core, compiled: oracle/extend_matrices.c (plain C restatement of
    tools.extend_matrices, reference twin cpp/src/tools.cc:83-144) on the C2 and C4 shapes."""
    from oracle import c_oracle

    out = []
    rng = np.random.default_rng(6)
    for name, n, m, N, count in (("C2 biped LIPM n=3 m=1 N=16", 3, 1, 16, 4096),
                                 ("C4 random LTI n=12 m=6 N=64", 12, 6, 64, 16)):
        A = rng.standard_normal((count, n, n)) / np.sqrt(n) * 0.9
        B = rng.standard_normal((count, n, m))
        c_oracle.extend_matrices_batch(A, B, N)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            c_oracle.extend_matrices_batch(A, B, N)
            done += count
        secs = time.perf_counter() - t0
        out.append({"shape": name, "systems_per_s": done / secs, "cores": 1,
                    "GBps": fill_bytes(n, m, N, False) * done / secs / 1e9,
                    "sample": "%d systems in %.1f s" % (done, secs)})
    return out


def _cpu_worker(args):
    """One process of the multi-core baseline: its own formulation (horizon matrices from the
    oracle, this process never touches the GPU), its own slice of the instances."""
    seed, budget_s = args
    sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
    sys.path.insert(0, ROOT)
    from oracle import qp_oracle as orc
    import mpc_interface.tools as tools

    tools.extend_matrices = orc.extend_matrices
    work = build_workload(256, seed)
    rate, done, elapsed = cpu_baseline(work, budget_s)
    return done, elapsed


def cpu_share():
    """Cores this process may really use: its affinity mask, cut down to the cgroup's CPU quota
    when there is one (a container on a 256-core host is often given 16)."""
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], float(fields[1])
            else:
                quota = fields[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = float(g.read().split()[0])
            if quota not in ("max", "-1") and period > 0:
                share = min(share, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return share


def cpu_baseline_all_cores(budget_s=10.0):
    """The same oracle loop in P processes, P = the cores this process may run on (at most 64:
    beyond that the pool's start-up outweighs the 10 s sample)."""
    import multiprocessing as mp

    procs = max(1, min(64, cpu_share()))
    with mp.get_context("spawn").Pool(procs) as pool:
        results = pool.map(_cpu_worker, [(1000 + i, budget_s) for i in range(procs)])
    done = sum(r[0] for r in results)
    return sum(r[0] / r[1] for r in results), done, procs


# --------------------------------------------------------------------------
# sub-records (timed after the main region; never part of `value`)
# --------------------------------------------------------------------------
def _event_ms(torch, fn, reps, warm=3, settle_ms=30.0):
    """Average duration of ``fn`` over ``reps`` back-to-back calls, hipEvents on the launch stream,
    after ``warm`` calls and at least ``settle_ms`` of load (the clocks settle, tools/launch_series.py)."""
    fn()                          # (a first call may compile a kernel)
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(4):
            fn()
        n += 4
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


FILL_CASES = [  # name, n, m, N, ltv, systems per GPU (north star: global batch / 8 GPUs)
    ("C2 biped LIPM n=3 m=1 N=16", 3, 1, 16, False, None),           # None: --batch
    ("C2 biped LIPM n=3 m=1 N=16", 3, 1, 16, False, 8192),
    ("C4 random LTI n=12 m=6 N=64", 12, 6, 64, False, 1024),
    ("C5 LTV LIPM n=3 m=1 N=100", 3, 1, 100, True, 2048),
]


def fill_bytes(n, m, N, ltv):
    """Algorithmic bytes of K1 per system: written + read."""
    return 8 * (N * n * n + m * N * N * n) + 8 * (n * n + n * m) * (N if ltv else 1)


def fill_records(torch, engine, dev, batch, reps=30):
    """K1 alone (mpcasm_fill_su) on the north-star shapes: fraction of the HBM roofline."""
    out = []
    rng = np.random.default_rng(5)
    for name, n, m, N, ltv, systems in FILL_CASES:
        systems = systems or batch
        shapeA = (systems, N, n, n) if ltv else (systems, n, n)
        shapeB = (systems, N, n, m) if ltv else (systems, n, m)
        A = torch.as_tensor(rng.standard_normal(shapeA) / np.sqrt(n) * 0.9, device=dev)
        Bm = torch.as_tensor(rng.standard_normal(shapeB), device=dev)
        S = torch.empty((systems, N, n, n), dtype=torch.float64, device=dev)
        U = torch.empty((systems, m, N, N, n), dtype=torch.float64, device=dev)
        ms = _event_ms(torch, lambda: engine.fill_su(A, Bm, N, ltv=ltv, out=(S, U)), reps)
        nbytes = fill_bytes(n, m, N, ltv) * systems
        gbps = nbytes / (ms * 1e-3) / 1e9
        out.append({"kernel": "mpcasm_fill_su (K1 toeplitz_fill%s)" % (", LTV" if ltv else ""),
                    "shape": name, "systems": systems, "bound": "hbm",
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms,
                    "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBPS, "systems_per_s": systems / (ms * 1e-3)})
        del A, Bm, S, U
    torch.cuda.empty_cache()
    return out


def variant_records(torch, dev, batch, seed):
    """The same step on the other shapes SURVEY.md 8d names for C2: the 34-wide bucket (the walking
    phase with one previewed step: no=34, nc=72) and the reduced formulation of the north star
    ("3 costs, 2 box constraints": no terminal cost, no terminal box), both widths."""
    out = []
    for name, kw in (("34-wide bucket (one previewed step)", dict(step_times=(7, 15))),
                     ("reduced: 3 cost kinds, 2 boxes; 36-wide", dict(reduced=True)),
                     ("reduced: 3 cost kinds, 2 boxes; 34-wide", dict(reduced=True, step_times=(7, 15)))):
        work = build_workload(batch, seed, **kw)
        engine, form = work["engine"], work["form"]
        asm = engine.Assembler(form, batch=batch, device=dev, lti=["LIP"])
        asm.bind_lti("LIP", torch.as_tensor(work["A"], device=dev), torch.as_tensor(work["B"], device=dev))
        asm.set_param("cost", "track vel_x", "aim", work["aims"])
        given = torch.as_tensor(work["given"], device=dev)
        f = dict(dtype=torch.float64, device=dev)
        sets = [(torch.empty((batch, asm.no, asm.no), **f), torch.empty((batch, asm.no), **f),
                 torch.empty((batch, asm.nc, asm.no), **f), torch.empty((batch, asm.nc), **f))
                for _ in range(4)]
        k = [0]

        def step():
            asm.assemble(given, out=sets[k[0] % 4])
            k[0] += 1

        ms = _event_ms(torch, step, 200)
        nbytes = 8 * (asm.no * asm.no + asm.no + asm.nc * asm.no + asm.nc) \
            + 8 * (asm.ng + int(asm.params.shape[1]) + 12)
        gbps = nbytes * batch / (ms * 1e-3) / 1e9
        out.append({"workload": "C2 biped N=16, %s" % name, "no": asm.no, "nc": asm.nc,
                    "costs": len(form.goals), "boxes": len(form.constraint_boxes),
                    "batch_per_gpu": batch, "kernel": asm.last_kernel(), "avg_launch_ms": ms,
                    "assemblies_per_s": batch / (ms * 1e-3), "algorithmic_bytes_per_assembly": nbytes,
                    "achieved": gbps, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS})
        del asm, sets, given
    torch.cuda.empty_cache()
    return out


def f2_records(torch, dev, seed, batches=(4096, 65536)):
    """SURVEY.md 8 row f2, the step after the solve: every row of every definition (body.py:209-219) for a
    batch of C2 walkers straight from the sources -- no preview matrix in memory (mpcasm_preview_direct)."""
    from mpcasm import engine

    out = []
    for batch in batches:
        work = build_workload(batch, seed)
        asm = engine.Assembler(work["form"], batch=batch, device=dev)
        given = torch.as_tensor(work["given"], device=dev)
        optim = torch.as_tensor(np.random.default_rng(seed).normal(0, 0.5, [batch, asm.no]), device=dev)
        rows = asm.preview_rows(given, optim)
        ms = _event_ms(torch, lambda: asm.preview_rows(given, optim, out=rows), 200)
        nbytes = 8 * (asm.ng + asm.no + asm.plan.pmrows)
        gbps = nbytes * batch / (ms * 1e-3) / 1e9
        out.append({"workload": "C2 biped N=16: rows of all definitions from [given; optim], shared S, U",
                    "batch_per_gpu": batch, "rows": int(asm.plan.pmrows), "avg_launch_ms": ms,
                    "instances_per_s": batch / (ms * 1e-3), "algorithmic_bytes_per_instance": nbytes,
                    "achieved": gbps, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS})
        del asm, rows, given, optim
    torch.cuda.empty_cache()
    return out


def admm_record(torch, dev, seed, batch=4096):
    """SURVEY.md 8 row f3, the "or": the solve itself on the assembled QPs (mpcasm_admm: OSQP's ADMM iteration,
    biped_mpc_loop.py:60) -- a fleet of C2 walkers near their nominal state, factor + inverse per call and the
    time of one iteration of the whole batch."""
    from mpcasm import engine

    work = build_workload(batch, seed)
    asm = engine.Assembler(work["form"], batch=batch, device=dev)
    given = torch.as_tensor(work["given"] * 1e-2, device=dev)
    P, q, G, h = (t.clone() for t in asm.assemble(given))
    ms0 = _event_ms(torch, lambda: engine.admm(P, q, G, h, iters=0, rho=1.0, residuals=False), 10)
    ms100 = _event_ms(torch, lambda: engine.admm(P, q, G, h, iters=100, rho=1.0, residuals=False), 5)
    x, y, z, res = engine.admm(P, q, G, h, iters=400, rho=1.0)
    return {"workload": "C2 biped N=16 QPs (no=%d, nc=%d) from mpcasm_assemble, OSQP's ADMM iteration, rho = 1"
                        % (asm.no, asm.nc),
            "batch_per_gpu": batch, "factor_ms": ms0, "us_per_iteration": (ms100 - ms0) * 10.0,
            "qp_iterations_per_s": batch * 100 / ((ms100 - ms0) * 1e-3),
            "residuals_after_400": [float(res[:, 0].max()), float(res[:, 1].max())]}


F64_MFMA_PEAK_TFLOPS = 78.6    # dense fp64 matrix peak of gfx950 (v_mfma_f64_16x16x4_f64: 32 flop/clk/SIMD)


def c4_record(torch, dev, batch=8192, reps=3):
    """BASELINE config C4 (random LTI nx=12 nu=6 N=64: no=384, nc=1536) at its per-GPU batch in ONE
    call on the tiled kernel, a different system in every instance (horizon tables generated from
    (A, B): no S, U in memory, no workspace).  Every cost is the full horizon of one state: the kernel
    sums P along diagonals (scan form) and only has to write -- bound: HBM.  The Toeplitz form of the
    same plan (windows of the table multiplied on the fp64 matrix core) is timed beside it."""
    from mpcasm import capi, engine, problems

    nx, nu, N = 12, 6, 64
    need = 8 * batch * (384 * 384 + 384 + 1536 * 384 + 1536) * 1.05
    free = torch.cuda.mem_get_info(dev)[0]
    while need > 0.8 * free and batch > 256:
        batch //= 2
        need /= 2
    api = problems.load_api("mpc_interface")
    rng = np.random.default_rng(20262)
    form = problems.random_lti(api, rng, nx=nx, nu=nu, N=N)
    base = [problems.random_lti_matrices(rng, nx, nu) for _ in range(64)]
    scale = 1.0 - 0.05 * rng.random(batch)
    A = np.stack([base[b % 64][0] * scale[b] for b in range(batch)])
    Bm = np.stack([base[b % 64][1] * (2.0 - scale[b]) for b in range(batch)])
    asm = engine.Assembler(form, batch=batch, device=dev, lti=["plant"])
    asm.bind_lti("plant", torch.as_tensor(A, device=dev), torch.as_tensor(Bm, device=dev))
    asm.set_param("cost", "track s0", "weight", rng.uniform(0.1, 1.0, [batch, 1, 1]))
    given = torch.as_tensor(rng.normal(0, 0.3, [batch, form.given_len]), device=dev)
    ms = _event_ms(torch, lambda: asm.assemble(given), reps, warm=1, settle_ms=0.0)
    kernel = asm.last_kernel()
    no, nc = asm.no, asm.nc
    out_bytes = 8 * (no * no + no + nc * no + nc)
    in_bytes = 8 * (asm.ng + int(asm.params.shape[1]) + nx * nx + nx * nu)
    gbps = (out_bytes + in_bytes) * batch / (ms * 1e-3) / 1e9
    rec = {"workload": "C4: random LTI nx=12 nu=6 N=64, no=%d nc=%d, per-instance (A,B), weight, given; "
                       "B=%d in one call" % (no, nc, batch),
           "kernel": kernel, "batch_per_gpu": batch, "ms_per_call": ms, "assemblies_per_s": batch / (ms * 1e-3),
           "algorithmic_bytes_per_assembly": out_bytes + in_bytes, "bound": "hbm",
           "hbm": {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS},
           "workspace_bytes_per_instance": 8 * int(asm.plan.tiled["work"])}
    # the same form with its set-up as pre-pass kernels and scratch in HBM (MPCASM_OPT_PATH 1: round 4's first version)
    asm.set_option(capi.OPT_PATH, 1)
    ms1 = _event_ms(torch, lambda: asm.assemble(given), reps, warm=1, settle_ms=0.0)
    rec["with_pre_passes"] = {"kernel": asm.last_kernel(), "ms_per_call": ms1}
    # the same plan with its windows multiplied on the matrix core (MPCASM_OPT_PATH 4)
    asm.set_option(capi.OPT_PATH, 4)
    ms4 = _event_ms(torch, lambda: asm.assemble(given), 2, warm=1, settle_ms=0.0)
    stages = asm.plan.tiled["stages"]
    nb = -(-no // 128)
    blocks = nb * (nb + 1) // 2                     # (P symmetric: block pairs bi <= bj)
    p_stages = [(int(st[3]) >> 16, (int(st[3]) >> 8) & 0xFF) for st in stages if (int(st[3]) >> 8) & 1]
    roles = int(asm.plan.tiled["toeplitz"]) and all(fl & 16 for _, fl in p_stages)     # TS_FLAG_SAME
    mfma = int(sum((n * n) * 4 * 4 * (blocks - nb) + ((2 * n * n + n) * 4 if roles else n * n * 16) * nb
                   for n, _ in p_stages))
    tflops = mfma * 2048 * batch / (ms4 * 1e-3) / 1e12
    rec["toeplitz_form"] = {"kernel": asm.last_kernel(), "ms_per_call": ms4, "assemblies_per_s": batch / (ms4 * 1e-3),
                            "hbm_frac": (out_bytes + in_bytes) * batch / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "executed_mfma_per_assembly": mfma, "mfma_TFLOPs": tflops,
                            "mfma_frac": tflops / F64_MFMA_PEAK_TFLOPS}
    del asm
    torch.cuda.empty_cache()
    # one model for the whole batch (S, U of the formulation's own system read from memory, shared; weights and
    # `given` per instance): the shared-model form -- P as a weighted sum of per-weight Hessians computed once per
    # launch -- and, beside it, the general form on the same launch (every instance's tiles multiplied anew)
    shared = engine.Assembler(form, batch=batch, device=dev)
    shared.set_param("cost", "track s0", "weight", rng.uniform(0.1, 1.0, [batch, 1, 1]))
    ms_s = _event_ms(torch, lambda: shared.assemble(given), 2, warm=1, settle_ms=0.0)
    rec["shared_model"] = {"kernel": shared.last_kernel(), "ms_per_call": ms_s,
                           "assemblies_per_s": batch / (ms_s * 1e-3),
                           "hbm_frac": out_bytes * batch / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    shared.set_option(capi.OPT_PATH, 3)
    ms_g = _event_ms(torch, lambda: shared.assemble(given), 1, warm=1, settle_ms=0.0)
    rec["general_form"] = {"kernel": shared.last_kernel(), "ms_per_call": ms_g,
                           "hbm_frac": out_bytes * batch / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    del shared, given
    torch.cuda.empty_cache()
    return rec


def c5_record(torch, dev, batch=2048, reps=10):
    """BASELINE config C5 as an ASSEMBLY (problems.lipm_ltv: LTV LIPM, N=100, two axes: no=200, nc=404) at
    its per-GPU batch, per-step per-instance (A_k, B_k), on the sweep kernel -- no horizon matrix is
    formed; algorithmic bytes: the results + 8 N (n^2 + n m) of (A_k, B_k) + given + parameters."""
    from mpcasm import engine, problems

    N = 100
    api = problems.load_api("mpc_interface")
    rng = np.random.default_rng(20263)
    form = problems.lipm_ltv(api, N=N)
    first = [problems.ltv_lipm_steps(api, N=N, theta=float(t)) for t in rng.uniform(0, 2 * np.pi, 8)]
    A = torch.as_tensor(np.stack([first[i % 8][0] * (1.0 - 1e-3 * (i // 8) / max(batch // 8, 1))
                                  for i in range(batch)]), device=dev)
    Bm = torch.as_tensor(np.stack([first[i % 8][1] * (1.0 + 1e-3 * (i // 8) / max(batch // 8, 1))
                                   for i in range(batch)]), device=dev)
    asm = engine.Assembler(form, batch=batch, device=dev, ltv=["LIP"])
    asm.bind_ltv("LIP", A, Bm)
    given = torch.as_tensor(rng.normal(0, 0.05, [batch, form.given_len]), device=dev)
    ms = _event_ms(torch, lambda: asm.assemble(given), reps)
    no, nc = asm.no, asm.nc
    nbytes = 8 * (no * no + no + nc * no + nc) + 8 * (N * 12 + asm.ng + int(asm.params.shape[1]))
    gbps = nbytes * batch / (ms * 1e-3) / 1e9
    rec = {"workload": "C5: LTV LIPM (dP->CCC) N=100, 2 axes, no=%d nc=%d, per-step per-instance (A_k,B_k); B=%d"
                       % (no, nc, batch),
           "kernel": asm.last_kernel(), "batch_per_gpu": batch, "ms_per_call": ms,
           "assemblies_per_s": batch / (ms * 1e-3), "algorithmic_bytes_per_assembly": nbytes, "bound": "hbm",
           "hbm": {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS}}
    del asm, given, A, Bm
    torch.cuda.empty_cache()
    return rec


def c3_record(torch, dev, batch=16384, sets=8, reps=10):
    """BASELINE config C3 (3-D LIPM N=32: no=96, nc=196) at B=16384, horizon matrices built on chip, into
    `sets` separately allocated sets of result buffers, all held at once (where P and G lie in memory moves
    this configuration: min / median / max over the sets)."""
    from mpcasm import engine, problems

    api = problems.load_api("mpc_interface")
    form = problems.lipm3d(api, N=32)
    rng = np.random.default_rng(20261)
    asm = engine.Assembler(form, batch=batch, device=dev, lti=["LIP"])
    given = torch.as_tensor(rng.normal(0, 0.05, [batch, form.given_len]), device=dev)
    f = dict(dtype=torch.float64, device=dev)
    no, nc = asm.no, asm.nc
    held = [(torch.empty((batch, no, no), **f), torch.empty((batch, no), **f),
             torch.empty((batch, nc, no), **f), torch.empty((batch, nc), **f)) for _ in range(sets)]
    times = [_event_ms(torch, lambda o=o: asm.assemble(given, out=o), reps, settle_ms=10.0) for o in held]
    nbytes = 8 * (no * no + no + nc * no + nc) + 8 * (asm.ng + int(asm.params.shape[1]) + 12)
    med = float(np.median(times))
    rec = {"workload": "C3: 3-D LIPM N=32, no=%d nc=%d, horizon matrices on chip; B=%d" % (no, nc, batch),
           "kernel": asm.last_kernel(), "batch_per_gpu": batch, "result_buffer_sets": sets,
           "ms_per_call": {"min": min(times), "median": med, "max": max(times)},
           "assemblies_per_s": batch / (med * 1e-3), "algorithmic_bytes_per_assembly": nbytes, "bound": "hbm",
           "hbm_frac": {"best": nbytes * batch / (min(times) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "median": nbytes * batch / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "worst": nbytes * batch / (max(times) * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
    del asm, held, given
    torch.cuda.empty_cache()
    return rec


def c1_record(ticks=64):
    """BASELINE config C1: the walking loop's tick for ONE instance through the drop-in API -- update(),
    generate_all_qp_matrices(given), results on the host (biped_mpc_loop.py:50-56) -- beside the same tick
    with the oracle (numpy on one host core) in place of the kernels."""
    from mpcasm import problems
    from oracle import qp_oracle as orc

    api = problems.load_api("mpc_interface")
    conf = problems.BipedConfig(step_samples=8)
    form = problems.biped(api, conf)
    n = conf.step_samples
    rng = np.random.default_rng(0)

    def tick(k, solve):
        phi = k % n
        form.update(step_times=np.array([(i + 1) * n - 1 - phi for i in range(conf.num_steps)]), step_count=k // n)
        return solve(rng.normal(0, 0.1, [form.given_len, 1]))

    out = {}
    for name, solve in (("tick_ms", form.generate_all_qp_matrices),
                        ("oracle_tick_ms", lambda given: orc.assemble(form, given))):
        for k in range(2 * n):
            tick(k, solve)
        t0 = time.perf_counter()
        for k in range(ticks):
            tick(k, solve)
        out[name] = (time.perf_counter() - t0) / ticks * 1e3
    out["workload"] = "C1: biped N=16, one instance, update() + generate_all_qp_matrices(), results on the host"
    return out


def _sig(x, digits=4):
    """Numbers of the one-line record: a few significant digits."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(rec):
    """The ONE line of the contract, short enough to survive a 2000-character tail: every key of the
    contract with its full meaning, the sub-records as numbers under short keys; the long form of the
    same record (workload descriptions, clocks, byte counts) goes to bench_full.json and stderr."""
    def short_kernel(name):
        return str(name).split(" ")[0]

    r = rec["roofline"]
    out = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "steps", "warmup",
                               "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c = rec["config"]
    out["config"] = {"workload": "C2 biped LIPM N=16 no=%d nc=%d, per-instance (A,B)/given/aim, %s"
                                 % (c["no"], c["nc"], "K1 on chip" if c["fused"] else "fill+assemble"),
                     "batch_per_gpu": c["batch_per_gpu"], "global_batch": c["global_batch"],
                     "streams": c["launch_streams"]}
    out["roofline"] = {"kernel": short_kernel(r["kernel_name"]), "bound": r["bound"], "achieved": r["achieved"],
                       "peak": r["peak"], "unit": r["unit"], "frac": r["frac"], "traffic": r["traffic"],
                       "bytes": r["algorithmic_bytes_per_launch"], "avg_launch_ms": r["avg_launch_ms"]}
    if "cpu_baseline" in rec:
        b = rec["cpu_baseline"]
        out["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                               "sample": b["sample_short"], "one_core": b["single_core_value"]}
        if "numpy_oracle" in b:   # (the interpreted port beside the compiled one)
            out["cpu_baseline"]["numpy"] = {"value": b["numpy_oracle"]["value"],
                                            "one_core": b["numpy_oracle"]["single_core_value"]}
    out["devices"] = rec["devices"]
    if "runtime_settings" in rec:
        out["kernarg_dev"] = rec["runtime_settings"]["HIP_FORCE_DEV_KERNARG"]
    if "gather" in rec:
        g = rec["gather"]
        out["gather"] = {"backend": g["backend"], "ms": g["ms"], "GBps": g["GBps_received_per_gpu"],
                         "instances_per_gpu_after": g["instances_per_gpu_after"]}
    if "fill" in rec:        # K1 alone: fraction of HBM at (shape @ systems)
        out["fill"] = {"%s@%d" % (f["shape"].split(" ")[0], f["systems"]): f["frac"] for f in rec["fill"]}
    if "variants" in rec:    # the other C2 shapes: fraction of HBM
        out["variants"] = {("w%d%s" % (v["no"], "r" if "reduced" in v["workload"] else "")): v["frac"]
                           for v in rec["variants"]}
    for key in ("c1", "c3", "c4", "c5", "admm", "f2", "extra"):
        v = rec.get(key)
        if v is None:
            continue
        if isinstance(v, dict) and "error" in v:
            out[key] = {"error": v["error"][:60]}
        elif key == "c1":
            out[key] = {"tick_ms": v["tick_ms"], "oracle_tick_ms": v["oracle_tick_ms"]}
        elif key == "c3":
            out[key] = {"B": v["batch_per_gpu"], "ms": [v["ms_per_call"][k] for k in ("min", "median", "max")],
                        "per_s": v["assemblies_per_s"], "frac": v["hbm_frac"]["median"]}
        elif key == "c4":
            out[key] = {"B": v["batch_per_gpu"], "kernel": short_kernel(v["kernel"]), "ms": v["ms_per_call"],
                        "per_s": v["assemblies_per_s"], "frac": v["hbm"]["frac"],
                        "mfma_form": {"ms": v["toeplitz_form"]["ms_per_call"],
                                      "mfma_frac": v["toeplitz_form"]["mfma_frac"]}}
            if "with_pre_passes" in v:
                out[key]["prepass_ms"] = v["with_pre_passes"]["ms_per_call"]
            if "shared_model" in v:     # (S, U shared by the batch: the shared-model form, the general form)
                out[key]["shared_ms"] = v["shared_model"]["ms_per_call"]
                out[key]["general_ms"] = v["general_form"]["ms_per_call"]
        elif key == "c5":
            out[key] = {"B": v["batch_per_gpu"], "kernel": short_kernel(v["kernel"]), "ms": v["ms_per_call"],
                        "per_s": v["assemblies_per_s"], "frac": v["hbm"]["frac"]}
        elif key == "admm":
            out[key] = {"B": v["batch_per_gpu"], "factor_ms": v["factor_ms"], "iter_us": v["us_per_iteration"]}
        elif key == "f2":
            out[key] = {"B%d" % f["batch_per_gpu"]: {"ms": f["avg_launch_ms"], "frac": f["frac"]} for f in v}
        elif key == "extra":
            out["B65536"] = {"ms": v["ms_per_step"], "per_s": v["assemblies_per_s"], "frac": v["frac"]}
    out["full"] = rec.get("full_record", None)
    return _sig(out)


def tiled(x, times):
    return np.concatenate([x] * times) if times > 1 else x


# --------------------------------------------------------------------------
# one rank
# --------------------------------------------------------------------------
def run_stub(args, world, rank):
    """TEST ONLY: the launcher / sharding / gather / record plumbing without a GPU."""
    import torch
    import torch.distributed as dist

    from mpcasm import dist as mdist

    if rank == args.stub_fail_rank:
        sys.exit(3)
    use_dist = world > 1 or args.dist_at_world_1
    if use_dist:
        _rendezvous_port(world)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    B, no, nc = args.batch, 36, 76
    lo, hi = mdist.shard_bounds(world * B, world, rank)
    P = torch.full((hi - lo, no, no), float(rank), dtype=torch.float64)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        P.add_(0.0)
    elapsed = time.perf_counter() - t0
    record = {"metric": METRIC, "unit": "assemblies/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "stub": True,
              "n_ranks_seen": dist.get_world_size() if use_dist else 1,
              "config": {"batch_per_gpu": B, "global_batch": B * world}}
    if use_dist:
        elapsed = mdist.max_over_ranks(elapsed)
        allP = mdist.gather_batch(P, world * B)
        assert allP.shape[0] == world * B
        assert all(float(allP[r * B, 0, 0]) == float(r) for r in range(world))
        record["gather"] = {"instances": int(allP.shape[0])}
        dist.destroy_process_group()
    record["value"] = world * B * args.steps / elapsed
    record["ms_per_step"] = elapsed / args.steps * 1e3
    if rank == 0:
        print(json.dumps(record), flush=True)


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world                   # (under a launcher: the ranks it started)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.stub_kernels:
        return run_stub(args, world, rank)

    import torch

    from mpcasm import dist as mdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the assembly path has no CPU fallback)")
    if args.backend == "gloo":      # rehearsal: the ranks may share a GPU (RCCL would refuse that)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dist_at_world_1:
        import torch.distributed as dist

        _rendezvous_port(world)
        kw = {}
        if args.backend == "nccl":
            # (the rank's own GPU by name: RCCL otherwise guesses it from the global rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    B = args.batch
    lo, hi = mdist.shard_bounds(world * B, world, rank)   # this rank's slice of the global batch
    assert hi - lo == B
    work = build_workload(B, 20260 + rank)                # its own instances
    engine, form, N = work["engine"], work["form"], work["N"]
    dev = torch.device("cuda", local_rank)
    A = torch.as_tensor(work["A"], device=dev)
    Bm = torch.as_tensor(work["B"], device=dev)
    given = torch.as_tensor(work["given"], device=dev)
    fused = not args.two_kernels

    def make_assembler(batch, A, Bm, aims):
        if fused:
            # K1 inside the assembly: per-instance (A, B) in, the kernel builds what it needs of
            # S, U in LDS (SURVEY.md section 8d counts exactly these bytes for an assembly)
            asm = engine.Assembler(form, batch=batch, device=dev, lti=["LIP"])
            asm.bind_lti("LIP", A, Bm)
            SU = None
        else:
            S = torch.empty((batch, N, 3, 3), dtype=torch.float64, device=dev)
            U = torch.empty((batch, 1, N, N, 3), dtype=torch.float64, device=dev)
            asm = engine.Assembler(form, batch=batch, device=dev)
            asm.bind_source(("LIP", 0), U[:, 0])
            asm.bind_source(("LIP", 1), S)
            SU = (S, U)
        asm.set_param("cost", "track vel_x", "aim", aims)
        return asm, SU

    def output_sets(asm, batch, count):
        f = dict(dtype=torch.float64, device=dev)
        return [(torch.empty((batch, asm.no, asm.no), **f), torch.empty((batch, asm.no), **f),
                 torch.empty((batch, asm.nc, asm.no), **f), torch.empty((batch, asm.nc), **f))
                for _ in range(max(1, count))]

    asm, SU = make_assembler(B, A, Bm, work["aims"])
    # (--two-kernels: the steps share the S, U buffer between fill and assembly -- one stream)
    nstreams = max(1, args.streams) if fused else 1
    # two launches in flight never write the same output set: at least twice as many sets as streams
    outs = output_sets(asm, B, max(args.rotate, 2 * nstreams if nstreams > 1 else 1))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nstreams - 1)]

    def step(k, stream=None):
        if not fused:
            engine.fill_su(A, Bm, N, out=SU, stream=stream)
        return asm.assemble(given, out=outs[k % len(outs)], stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:       # (one rank alone: nothing to wait for behind the first synchronize)
            dist.barrier()
            torch.cuda.synchronize()

    # clocks first (tools/launch_series.py), then the W warm-up steps of the contract
    step(0)                       # (the first call compiles the kernel for the plan)
    torch.cuda.synchronize()
    settle_steps, t_settle = 0, time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(16):
            step(settle_steps, streams[settle_steps % nstreams])
            settle_steps += 1
        torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k, streams[k % nstreams])
    sync_all()

    # timed region: exactly K steps, dealt to the launch streams in turn, between two barrier +
    # synchronize pairs and nothing else -- no event, no wait between the streams (the device is idle
    # at its start and synchronised as a whole at its end): with the driver's 20 steps the two event
    # records and the two cross-stream waits of rounds 2-3 were a seventh of the region.
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, streams[k % nstreams])
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = mdist.max_over_ranks(elapsed, device=dev)
    # the same K steps once more with a pair of hipEvents around them (the first on stream 0, the
    # others wait for it; the second on stream 0 after it has waited for the others): the device's own
    # clock beside the wall clock, `timed_region.ms_per_step_hipEvents`; not what `value` comes from
    e_begin, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_begin.record(streams[0])
    for s in streams[1:]:
        s.wait_event(e_begin)
    for k in range(args.steps):
        step(k, streams[k % nstreams])
    for s in streams[1:]:
        streams[0].wait_stream(s)
    e_end.record(streams[0])
    sync_all()
    region_ms = e_begin.elapsed_time(e_end) / args.steps

    # the dominant kernel by itself, for the roofline record: the same steps once more on ONE
    # stream (K1-fused: a step IS one kernel, so event time / steps = average launch duration,
    # launch gaps included -- the conservative reading; rocprofv3's kernel time of the same
    # command with --streams 1 is what it is compared with); --two-kernels adds a pair of events
    # around the fill of every 8th step.
    k_steps = max(min(args.steps, 1000), 200)      # (a short run of the driver's: still 200 launches behind the kernel's figure)
    k_begin, k_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fill_ev = []
    k_begin.record()
    for k in range(k_steps):
        if not fused:
            if k % 8 == 0:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record()
                engine.fill_su(A, Bm, N, out=SU)
                pair[1].record()
                fill_ev.append(pair)
            else:
                engine.fill_su(A, Bm, N, out=SU)
        asm.assemble(given, out=outs[k % len(outs)])
    k_end.record()
    sync_all()
    step_ms = k_begin.elapsed_time(k_end) / k_steps
    fill_ms = float(np.mean([a.elapsed_time(b) for a, b in fill_ev])) if fill_ev else 0.0
    asm_ms = step_ms - fill_ms

    no, ng, nc = asm.no, asm.ng, asm.nc
    nparams = int(asm.params.shape[1])
    bytes_fill = fill_bytes(3, 1, N, False)
    bytes_out = 8 * (no * no + no + nc * no + nc)
    bytes_asm = bytes_out + 8 * (ng + nparams) + (8 * (9 + 3) if fused else 8 * (N * 9 + N * N * 3))
    value = world * B * args.steps / elapsed

    # dominant kernel = the assembly (K2-K4, K1 inside when fused); HBM-bound
    achieved = bytes_asm * B / (asm_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in separate runs, tools/summarize_profile.py); only
    # quoted when it was collected at this batch size
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("batch_per_gpu") == B and pmc.get("k1_fused", False) == fused:
            traffic = pmc["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    # (which kernel really ran: the per-plan compiled one falls back to the ahead-of-time kernel
    # when libhiprtc.so is missing or the compilation fails)
    kernel_name = "mpcasm_assemble -> %s: %sK2 compose + K3 hessian_mfma + K4 constraint_stack in one " \
                  "launch" % (asm.last_kernel(), "K1 horizon tables + " if fused else "")
    ranks_seen = dist.get_world_size() if dist is not None else 1
    devices = ["r%d cuda:%d %s" % (rank, local_rank, torch.cuda.get_device_name(local_rank).replace("AMD Instinct ", ""))]
    if dist is not None:
        names = [None] * ranks_seen
        dist.all_gather_object(names, devices[0])
        devices = names
    timed_ms = elapsed * 1e3
    if timed_ms < 5.0 and rank == 0:
        print("bench.py: the timed region is only %.2f ms (%d steps): a few launches of jitter move "
              "`value` by several per cent -- use --steps %d or more for a steady number"
              % (timed_ms, args.steps, int(np.ceil(50.0 / max(timed_ms / args.steps, 1e-6)))),
              file=sys.stderr)
    record = {
        "metric": METRIC,
        "value": value,
        "unit": "assemblies/s",
        "n_gpus": world,
        "n_ranks_seen": ranks_seen,
        "devices": devices,
        "runtime_settings": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
        "steps": args.steps,
        "warmup": args.warmup,
        "timed_region_ms": timed_ms,
        "settle": {"ms": args.settle_ms, "untimed_steps_before_warmup": settle_steps},
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "C2: biped LIPM (J->CCC) N=16, 2 axes, 6 costs, 3 boxes; no=%d nc=%d ng=%d; "
                        "per-instance (A,B), given, velocity aim; B=%d per GPU" % (no, nc, ng, B),
            "batch_per_gpu": B,
            "global_batch": B * world,
            "no": no, "nc": nc, "fused": bool(fused),
            "horizon": N,
            "output_buffer_sets": len(outs),
            "launch_streams": nstreams,
            "step": "mpcasm_assemble, horizon matrices built on chip from per-instance (A,B) "
                    "(K1 fused)" if fused else "mpcasm_fill_su + mpcasm_assemble",
        },
        "roofline": {
            "kernel": kernel_name,
            "kernel_name": asm.last_kernel(),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": "profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of this command "
                              "at this batch, committed with the round's profiles; null at any other "
                              "batch) -- not measured in this run",
            "algorithmic_bytes_per_launch": bytes_asm * B,
            "algorithmic_bytes_per_assembly": bytes_asm,
            "avg_launch_ms": asm_ms,
            "clock": "hipEvents on ONE launch stream around %d launches of the same steps, right after the "
                     "timed region (launch gaps included); the timed region itself deals its launches to "
                     "%d stream(s), where consecutive launches overlap" % (k_steps, nstreams),
        },
        "timed_region": {"launch_streams": nstreams, "ms_per_step_hipEvents": region_ms,
                         "one_stream_ms_per_step": step_ms},
        "hbm_GBps_end_to_end": (bytes_asm + (0 if fused else bytes_fill)) * value / 1e9,
    }
    if not fused:
        record["fill_in_step"] = {
            "kernel": "mpcasm_fill_su (K1 toeplitz_fill)",
            "algorithmic_bytes_per_system": bytes_fill,
            "avg_launch_ms": fill_ms,
            "achieved_GBps": bytes_fill * B / (fill_ms * 1e-3) / 1e9,
        }

    if not args.no_extras:
        # optional final gather of the assembled QPs: all-gather over RCCL/xGMI
        if dist is not None:
            P, q, G, h = outs[(args.steps - 1) % len(outs)]
            gathered = [None]

            def gather():
                gathered[0] = [mdist.gather_batch(t, world * B) for t in (P, q, G, h)]

            for _ in range(2):
                gather()
            sync_all()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                gather()
            sync_all()
            g_ms = mdist.max_over_ranks((time.perf_counter() - t0) / reps * 1e3, device=dev)
            recv = bytes_out * B * max(world - 1, 1)   # (one rank: its own slice, device to device)
            record["gather"] = {
                "what": "all-gather of P,q,G,h of every rank onto every rank (mpcasm.dist.gather_batch), "
                        "after the assembly; not part of `value`",
                "backend": args.backend, "ms": g_ms, "bytes_received_per_gpu": recv,
                "GBps_received_per_gpu": recv / (g_ms * 1e-3) / 1e9,
                "instances_per_gpu_after": int(gathered[0][0].shape[0]),
            }
            del gathered
            torch.cuda.empty_cache()
        # the other C2 shapes (34-wide bucket, the north star's reduced formulation)
        record["variants"] = variant_records(torch, dev, B, 20260 + rank)
        # K1 alone on the north-star shapes
        record["fill"] = fill_records(torch, engine, dev, B)
        # f2: the step after the solve, on the device
        try:
            record["f2"] = f2_records(torch, dev, 20260 + rank)
        except Exception as exc:        # (never at the cost of the main line)
            record["f2"] = {"error": repr(exc)}
        # the other BASELINE configurations: C1 (the drop-in tick of one instance), C3, C4, C5 at their
        # per-GPU batches (never at the cost of the main line).  They are figures of ONE GPU: a run on
        # several ranks does not repeat them on every rank (tens of GB of results each, minutes of
        # wall clock between the timed region and the line the driver waits for)
        single = world == 1
        for key, make in (("c1", lambda: c1_record()), ("c3", lambda: c3_record(torch, dev)),
                          ("c4", lambda: c4_record(torch, dev)), ("c5", lambda: c5_record(torch, dev)),
                          ("admm", lambda: admm_record(torch, dev, 20260))):
            if not single:
                continue
            try:
                record[key] = make()
            except Exception as exc:
                record[key] = {"error": repr(exc)}
        # the same step at B=65536: 2.2 GB of outputs per step, no cache can hold it
        big = 65536
        times = (big + B - 1) // B
        if big > B and single:
            Ab = torch.as_tensor(tiled(work["A"], times)[:big], device=dev)
            Bb = torch.as_tensor(tiled(work["B"], times)[:big], device=dev)
            gb = torch.as_tensor(tiled(work["given"], times)[:big], device=dev)
            asm_big, SU_big = make_assembler(big, Ab, Bb, tiled(work["aims"], times)[:big])
            out_big = output_sets(asm_big, big, 1)[0]

            def big_step():
                if not fused:
                    engine.fill_su(Ab, Bb, N, out=SU_big)
                asm_big.assemble(gb, out=out_big)

            ms = _event_ms(torch, big_step, 20)
            gbps = (bytes_asm + (0 if fused else bytes_fill)) * big / (ms * 1e-3) / 1e9
            record["extra"] = {
                "what": "the same step at B=%d per GPU (outputs %.2f GB per step: HBM traffic for sure)"
                        % (big, bytes_out * big / 1e9),
                "batch_per_gpu": big, "ms_per_step": ms, "assemblies_per_s": big / (ms * 1e-3),
                "achieved": gbps, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            }
            del asm_big, SU_big, out_big, Ab, Bb, gb
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rate1, count1, secs1 = cpu_baseline(work, 5.0)
        rate, count, procs = cpu_baseline_all_cores(8.0)
        try:
            compiled = cpu_fill_compiled()
        except (OSError, RuntimeError) as exc:       # (oracle/liboracle.so not built)
            compiled = "unavailable: %s" % exc
        numpy_line = {"value": rate, "cores": procs, "single_core_value": rate1,
                      "sample": "%d assemblies of the same workload in 8 s on %d processes (one formulation "
                                "each): oracle/qp_oracle.py (extend_matrices + preview matrices + all QP "
                                "blocks per instance; the per-tick update() callback of the reference's "
                                "979/s figure is not in it); one process alone: %d in %.1f s; host has %d cores"
                                % (count, procs, count1, secs1, os.cpu_count() or 0)}
        try:
            # the stronger baseline: the same path compiled (oracle/assemble_port.c), one thread per core
            c1, n1 = cpu_baseline_compiled(work, 1, 3.0)
            cp, np_ = cpu_baseline_compiled(work, procs, 5.0)
            record["cpu_baseline"] = {
                "value": cp, "unit": "assemblies/s", "cores": procs, "kind": "port",
                "single_core_value": c1,
                "sample_short": "oracle/assemble_port.c (C port, gcc -O3): %d assemblies in 5 s on %d threads"
                                % (np_, procs),
                "sample": "%d assemblies of the same workload (per-instance (A, B), given, aim) in 5 s on %d threads: "
                          "extend_matrices + dense preview matrices + every limit and cost per instance, the loop "
                          "structure of body.py:142-348" % (np_, procs),
                "numpy_oracle": numpy_line, "compiled_fill_single_thread": compiled}
        except (OSError, RuntimeError, AttributeError) as exc:   # (liboracle.so not built, or an old one)
            record["cpu_baseline"] = {
                "value": rate, "unit": "assemblies/s", "cores": procs, "kind": "port",
                "single_core_value": rate1,
                "sample_short": "oracle (numpy port of the path), %d assemblies in 8 s on %d procs; 1 proc: %d in %.0f s"
                                % (count, procs, count1, secs1),
                "sample": numpy_line["sample"], "compiled_port": "unavailable: %s" % exc,
                "compiled_fill_single_thread": compiled}
    if rank == 0:
        # the long form beside the line: a file (gpurun_out/ travels back from a GPU box) and stderr
        full_dir = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(full_dir, exist_ok=True)
            # (the complete run keeps its name; a run with --no-extras does not overwrite it)
            name = "bench_full.json" if not args.no_extras else "bench_full_no_extras.json"
            with open(os.path.join(full_dir, name), "w") as f:
                json.dump(record, f, indent=1)
            record["full_record"] = "gpurun_out/" + name
        except OSError:
            record["full_record"] = "stderr"
        print(json.dumps(record), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(record)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus is None:
            args.gpus = 1
        if args.gpus > 1:
            sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
