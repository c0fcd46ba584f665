#!/usr/bin/env python3
"""Benchmark of the QP-assembly hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): the biped-LIPM walking
formulation, N=16, 2 axes, 6 costs, 3 boxes (36 unknowns, 76 inequality rows in
the headline phase), B independent instances per GPU in fp64, synthetic inputs
resident in HBM.  One *step* is one pass of the hot path over one batch:

    K1  mpcasm_fill_su     per-instance (A, B) -> horizon matrices S, U
    K2-K4 mpcasm_assemble  S, U, given, parameters -> P, q, G, h  for every instance

Prints ONE JSON line (rank 0): whole-job assemblies/s, the roofline object of the
dominant kernel (hipEvent-timed inside the timed region) and, at N=1, the CPU
baseline (the numpy oracle -- a port of the reference algorithm -- timed on this
box's host cores for a bounded sample of the same workload).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); the batch
is sharded with no collective on the data path (weak scaling: B per GPU is fixed);
the only collectives are the barriers around the timed region and the MAX of the
elapsed time.
"""
import argparse
import json
import os
import sys
import time

# the CPU baseline is quoted for one core: keep BLAS single-threaded
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (os.path.join(ROOT, "mpc-interface_amd"), ROOT):
    if path not in sys.path:
        sys.path.insert(0, path)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)


def build_workload(batch, seed):
    """The biped formulation in its 36-wide phase + per-instance synthetic inputs."""
    from mpcasm import engine, problems

    api = problems.load_api("mpc_interface")
    conf = problems.BipedConfig(step_samples=8)               # N = 16
    form = problems.biped(api, conf)
    form.update(step_times=np.array([6, 14]), step_count=0)   # phase phi=1: no=36, nc=76
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


def cpu_baseline_all_cores(budget_s=10.0):
    """The same oracle loop in P processes, P = this box's CPU share (at most 16)."""
    import multiprocessing as mp

    procs = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity")
                       else (os.cpu_count() or 1)))
    with mp.get_context("spawn").Pool(procs) as pool:
        results = pool.map(_cpu_worker, [(1000 + i, budget_s) for i in range(procs)])
    done = sum(r[0] for r in results)
    return sum(r[0] / r[1] for r in results), done, procs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-kernels", action="store_true",
                    help="step = mpcasm_fill_su (S, U through HBM) + mpcasm_assemble instead of the "
                         "default single launch that builds the horizon matrices on chip")
    ap.add_argument("--fused", action="store_true", help="(the default; kept for scripts)")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket the calls of every n-th timed step with hipEvents")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the assembly path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    B = args.batch
    work = build_workload(B, 20260 + rank)          # each rank owns its own shard of instances
    engine, form, N = work["engine"], work["form"], work["N"]
    dev = torch.device("cuda", local_rank)
    A = torch.as_tensor(work["A"], device=dev)
    Bm = torch.as_tensor(work["B"], device=dev)
    given = torch.as_tensor(work["given"], device=dev)
    fused = not args.two_kernels
    if fused:
        # K1 inside the assembly: per-instance (A, B) in, the kernel builds what it needs of
        # S, U in LDS (SURVEY.md section 8d counts exactly these bytes for an assembly)
        asm = engine.Assembler(form, batch=B, device=dev, lti=["LIP"])
        asm.bind_lti("LIP", A, Bm)
    else:
        S = torch.empty((B, N, 3, 3), dtype=torch.float64, device=dev)
        U = torch.empty((B, 1, N, N, 3), dtype=torch.float64, device=dev)
        asm = engine.Assembler(form, batch=B, device=dev)
        asm.bind_source(("LIP", 0), U[:, 0])
        asm.bind_source(("LIP", 1), S)
    asm.set_param("cost", "track vel_x", "aim", work["aims"])

    def step():
        if not fused:
            engine.fill_su(A, Bm, N, out=(S, U))
        return asm.assemble(given)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()

    # timed region: exactly K steps; hipEvents (on the launch stream) bracket the C-ABI
    # calls of every EVERY-th step, so that the kernels' own durations come from the same
    # run without an event pair between every two launches
    sampled = range(0, args.steps, args.event_every)
    ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for k in sampled}
    t0 = time.perf_counter()
    for k in range(args.steps):
        e = ev.get(k)
        if e:
            e[0].record()
        if not fused:
            engine.fill_su(A, Bm, N, out=(S, U))
        if e:
            e[1].record()
        asm.assemble(given)
        if e:
            e[2].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fill_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev.values()]))
    asm_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev.values()]))

    no, ng, nc = asm.no, asm.ng, asm.nc
    nparams = int(asm.params.shape[1])
    bytes_fill = 8 * (N * 9 + N * N * 3) + 8 * (9 + 3)                    # written + read
    bytes_out = 8 * (no * no + no + nc * no + nc)
    bytes_asm = bytes_out + 8 * (ng + nparams) + (8 * (9 + 3) if fused else 8 * (N * 9 + N * N * 3))
    total = world * B * args.steps
    value = total / elapsed

    # dominant kernel = the assembly (K2-K4); HBM-bound
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
    record = {
        "metric": "QP assemblies/sec (P,q,G,h), biped N=16 batched",
        "value": value,
        "unit": "assemblies/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
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
            "horizon": N,
            "step": "mpcasm_assemble, horizon matrices built on chip from per-instance (A,B) "
                    "(K1 fused)" if fused else "mpcasm_fill_su + mpcasm_assemble",
        },
        "roofline": {
            "kernel": "mpcasm_assemble -> resident_assemble_kernel (%sK2 compose + K3 hessian_mfma "
                      "+ K4 constraint_stack fused in one persistent launch)"
                      % ("K1 horizon tables + " if fused else ""),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_asm * B,
            "algorithmic_bytes_per_assembly": bytes_asm,
            "avg_launch_ms": asm_ms,
        },
        "hbm_GBps_end_to_end": (bytes_asm + (0 if fused else bytes_fill)) * value / 1e9,
    }
    if not fused:
        record["fill"] = {
            "kernel": "mpcasm_fill_su (K1 toeplitz_fill)",
            "algorithmic_bytes_per_system": bytes_fill,
            "avg_launch_ms": fill_ms,
            "achieved_GBps": bytes_fill * B / (fill_ms * 1e-3) / 1e9,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rate1, count1, secs1 = cpu_baseline(work, 6.0)
        rate, count, procs = cpu_baseline_all_cores(10.0)
        record["cpu_baseline"] = {
            "value": rate,
            "unit": "assemblies/s",
            "cores": procs,
            "kind": "port",
            "single_core_value": rate1,
            "sample": "%d assemblies of the same workload in 10 s on %d processes (one formulation "
                      "each): oracle/qp_oracle.py (extend_matrices + preview matrices + all QP "
                      "blocks per instance); one process alone: %d in %.1f s; host has %d cores"
                      % (count, procs, count1, secs1, os.cpu_count() or 0),
        }
    if rank == 0:
        print(json.dumps(record))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
