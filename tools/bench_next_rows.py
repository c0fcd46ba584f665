#!/usr/bin/env python3
"""Rates of the rows SURVEY.md section 8 marks "next" (f1-f4) on one GPU: the walker fleet's
ticks, the batched preview, the CSC hand-off and the box transforms, on the C2 biped."""
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import engine, problems  # noqa: E402
from mpcasm.boxes import BoxBatch  # noqa: E402
from mpcasm.walkers import WalkerFleet  # noqa: E402


def timed(fn, reps, warm=3, settle_ms=30.0):
    fn()
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < settle_ms:   # (the clocks settle under load)
        fn()
        n += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    work = bench.build_workload(B, 3)
    form = work["form"]
    given = torch.as_tensor(work["given"], device="cuda")
    asm = engine.Assembler(form, batch=B)
    P, q, G, h = asm.assemble(given)

    # f1: a fleet of walkers in all phases of the step cycle, one tick = clocks, step
    # indicator matrices, stepping centres, one assembly per structure bucket
    fleet = WalkerFleet(B, conf=problems.BipedConfig(step_samples=8))
    g_fleet = torch.zeros((B, fleet.given_len), dtype=torch.float64, device="cuda")
    # (the fleet keeps the inputs of the 2 * step_samples ticks of its cycle on the device: warm
    # all of them up, the rate is that of the steady state)
    t = timed(lambda: fleet.tick(g_fleet), 64, warm=2 * fleet.conf.step_samples + 4)
    print("f1 walker fleet   %6d walkers  %8.3f ms per tick  %10.0f walker-ticks/s" % (B, t * 1e3, B / t))
    # ... the same ticks replayed from hipGraphs (one per place in the step cycle)
    fleet = WalkerFleet(B, conf=problems.BipedConfig(step_samples=8), graphs=True)
    t = timed(lambda: fleet.tick(g_fleet), 64, warm=4 * fleet.conf.step_samples + 4)
    print("   from hipGraphs %6d walkers  %8.3f ms per tick  %10.0f walker-ticks/s" % (B, t * 1e3, B / t))
    fleet1 = WalkerFleet(B, conf=problems.BipedConfig(step_samples=8), graphs=True, side_by_side=True)
    t = timed(lambda: fleet1.tick(g_fleet), 64, warm=4 * fleet.conf.step_samples + 4)
    print("   from hipGraphs, the buckets side by side (branches)  %8.3f ms per tick  %10.0f walker-ticks/s" % (t * 1e3, B / t))
    del fleet1
    # ... with the walkers' states written straight into the fleet's own buffer (no copy per tick)
    g_own = fleet.given_buffer()
    t = timed(lambda: fleet.tick(g_own), 64, warm=8)
    print("   from hipGraphs, `given` in the fleet's buffer  %8.3f ms per tick  %10.0f walker-ticks/s" % (t * 1e3, B / t))

    # f2 (round 3): every row of every definition and every goal's distance straight from the
    # sources -- no preview matrix in memory (mpcasm_preview_direct, mpcasm_goal_distance)
    optim0 = torch.as_tensor(np.random.default_rng(1).normal(0, 0.5, [B, asm.no]), device="cuda")
    for label, kw in (("shared S, U", {}), ("tables from per-instance (A, B)", dict(lti=["LIP"]))):
        a2 = engine.Assembler(form, batch=B, **kw)
        if kw:
            a2.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
        rows = a2.preview_rows(given, optim0)
        t = timed(lambda: a2.preview_rows(given, optim0, out=rows), 100)
        td = timed(lambda: a2.goal_distance(form, rows), 100)
        algo = 8 * (asm.ng + asm.no + a2.plan.pmrows) + (8 * 12 if kw else 0)
        print("f2 rows of all definitions, %-32s %6d x %d rows  %8.3f ms  %10.0f instances/s  %6.0f GB/s "
              "(algorithmic: given + optim%s read, rows written)"
              % (label + ":", B, a2.plan.pmrows, t * 1e3, B / t, algo * B / t / 1e9, " + A, B" if kw else ""))
        print("   goal distances (%d goals)   %8.3f ms" % (len(form.goals), td * 1e3))
        tf = timed(lambda: a2.full_goal_distances(form, given, optim0), 100)
        print("   goal distances straight from the sources, rows never stored: %8.3f ms  %10.0f instances/s"
              % (tf * 1e3, B / tf))
        del a2
    # ... against round 2's two passes: the preview matrices through HBM, then a GEMV
    PM = asm.preview_matrices()
    optim = torch.zeros((B, asm.no), dtype=torch.float64, device="cuda")
    t = timed(lambda: asm.preview(PM, given, optim), 50)
    nbytes = PM.numel() * 8
    print("f2 preview        %6d x %d rows  %8.3f ms  %7.0f GB/s read" % (B, asm.plan.pmrows, t * 1e3, nbytes / t / 1e9))
    t = timed(lambda: asm.preview_matrices(), 20)
    print("   preview matrices (K2 alone)   %8.3f ms  %7.0f GB/s written" % (t * 1e3, nbytes / t / 1e9))

    # f3: data arrays of csc_matrix(P), csc_matrix(G) on the structural patterns
    for which, dense in (("P", P), ("G", G)):
        indptr, _ = asm.csc_pattern(which)
        t = timed(lambda: asm.export_csc(which), 50)
        print("f3 csc %s          nnz %5d of %6d  %8.3f ms  %7.0f GB/s written"
              % (which, indptr[-1], dense[0].numel(), t * 1e3, B * int(indptr[-1]) * 8 / t / 1e9))

    # f3, written by the assembly itself: horizon matrices built on chip, P (upper triangle) and G
    # leave as CSC data; against the dense assembly + the two gather passes on the same inputs
    A = torch.as_tensor(work["A"], device="cuda")
    Bm = torch.as_tensor(work["B"], device="cuda")
    dense = engine.Assembler(form, batch=B, lti=["LIP"])
    dense.bind_lti("LIP", A, Bm)
    sparse = engine.Assembler(form, batch=B, lti=["LIP"], csc="upper")
    sparse.bind_lti("LIP", A, Bm)
    c = sparse.csc

    def two_pass():
        dense.assemble(given)
        dense.export_csc("P", upper=True)
        dense.export_csc("G")

    t2 = timed(two_pass, 30)
    t1 = timed(lambda: sparse.assemble(given), 30)
    out_bytes = 8 * (c["pnnz"] + asm.no + c["gnnz"] + asm.nc)
    print("f3 csc in one pass  nnz P %d (upper) G %d: %d B per instance instead of %d"
          % (c["pnnz"], c["gnnz"], out_bytes, 8 * (asm.no * asm.no + asm.no + asm.nc * asm.no + asm.nc)))
    print("   dense assembly + 2 gathers %8.3f ms  %10.0f QPs/s;  CSC from the kernel %8.3f ms  %10.0f QPs/s  (%.0f GB/s written)"
          % (t2 * 1e3, B / t2, t1 * 1e3, B / t1, B * out_bytes / t1 / 1e9))

    # f3, the "or": the solve itself on the assembled QPs (mpcasm_admm: OSQP's ADMM iteration)
    Pd, qd, Gd, hd = asm.assemble(given)
    Pd, qd, Gd, hd = (t.clone() for t in (Pd, qd, Gd, hd))
    for iters in (0, 25, 100):
        t = timed(lambda: engine.admm(Pd, qd, Gd, hd, iters=iters, rho=1.0, residuals=False), 20)
        print("f3 admm %3d iterations, cold  %6d QPs (no %d nc %d)  %8.3f ms  %10.0f QPs/s"
              % (iters, B, asm.no, asm.nc, t * 1e3, B / t))
    xs, ys, zs, _ = engine.admm(Pd, qd, Gd, hd, iters=50, rho=1.0)
    t = timed(lambda: engine.admm(Pd, qd, Gd, hd, xs, ys, zs, iters=25, rho=1.0, residuals=False), 20)
    print("   warm, 25 iterations                       %8.3f ms  %10.0f QPs/s   (factor + inverse per call: the 0-iteration line)"
          % (t * 1e3, B / t))
    kinv = torch.empty((B, asm.no, asm.no), dtype=torch.float64, device="cuda")
    engine.admm(Pd, qd, Gd, hd, iters=0, rho=1.0, residuals=False, kinv=kinv)
    t = timed(lambda: engine.admm(Pd, qd, Gd, hd, xs, ys, zs, iters=25, rho=1.0, residuals=False, kinv=kinv,
                                  kinv_valid=True), 20)
    print("   warm, 25 iterations, K^-1 kept from the call before (P, G unchanged)  %8.3f ms  %10.0f QPs/s" % (t * 1e3, B / t))

    # f4: box transforms on the per-instance parameters
    box = BoxBatch(asm, form, "support_polygon")
    rot = torch.eye(2, dtype=torch.float64, device="cuda").repeat(B, 1, 1)
    shift = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    for name, fn in (("rotate", lambda: box.rotate_in_TS(rot)), ("translate", lambda: box.translate_in_TS(shift))):
        t = timed(fn, 50)
        print("f4 box %-9s  %6d boxes x %d facets  %8.3f ms  %10.0f boxes/s" % (name, B, box.nfacets, t * 1e3, B / t))


if __name__ == "__main__":
    main()
