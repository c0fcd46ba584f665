#!/usr/bin/env python3
"""Does the speed of a launch depend on WHERE its four result arrays lie?  One plan, one process: P, q, G,
h carved out of one big allocation at chosen byte offsets from each other, rounds interleaved.
ab_placement.py c3|c2 [batch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from mpcasm import engine, problems  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
api = problems.load_api("mpc_interface")
get_A, get_B, _ = api.tools.get_system_matrices("J->CCC")
if which == "c3":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    form = problems.lipm3d(api, N=32)
elif which == "c4":
    form = None
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    form = bench.build_workload(16, 1)["form"]
if which == "c4":                                # (random LTI nx=12 nu=6 N=64 on the tiled kernel)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rng = np.random.default_rng(20262)
    form = problems.random_lti(api, rng, nx=12, nu=6, N=64)
    base = [problems.random_lti_matrices(rng, 12, 6) for _ in range(64)]
    A = torch.as_tensor(np.stack([base[b % 64][0] for b in range(B)]), device="cuda")
    Bm = torch.as_tensor(np.stack([base[b % 64][1] for b in range(B)]), device="cuda")
    asm = engine.Assembler(form, batch=B, lti=["plant"])
    asm.bind_lti("plant", A, Bm)
else:
    taus = np.random.default_rng(1).uniform(0.08, 0.12, B)
    A = torch.as_tensor(np.stack([get_A(tau=t) for t in taus]), device="cuda")
    Bm = torch.as_tensor(np.stack([get_B(tau=t) for t in taus]), device="cuda")
    asm = engine.Assembler(form, batch=B, lti=["LIP"])
    asm.bind_lti("LIP", A, Bm)
given = torch.as_tensor(np.random.default_rng(0).normal(0, 0.1, [B, form.given_len]), device="cuda")
no, nc = asm.no, asm.nc
sizes = [B * no * no * 8, B * no * 8, B * nc * no * 8, B * nc * 8]
pool = torch.empty(sum(sizes) + ((5 << 30) if which != "c4" else (64 << 20)), dtype=torch.uint8, device="cuda")
base = (-pool.data_ptr()) % (2 << 20)           # a 2 MB boundary


def carve(gaps):
    """P, q, G, h behind each other, `gaps[k]` extra bytes in front of array k."""
    out, off = [], base
    for k, (nbytes, shape) in enumerate(zip(sizes, [(B, no, no), (B, no), (B, nc, no), (B, nc)])):
        off += gaps[k]
        out.append(pool[off:off + nbytes].view(torch.float64).view(shape))
        off += nbytes + (-nbytes) % 256
    return tuple(out)


def timed(out, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        asm.assemble(given, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = {"packed": (0, 0, 0, 0)}
for g in (256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20):
    cases["G +%d" % g] = (0, 0, g, 0)
for g in (8, 16, 32, 64, 128, 256, 512, 1024, 1536, 2048) if which != "c4" else ():
    cases["G +%d MB" % g] = (0, 0, g << 20, 0)
    cases["P +%d MB" % g] = (g << 20, 0, 0, 0)
cases["q, h +8 MB"] = (0, 8 << 20, 0, 8 << 20)
cases["all +4096"] = (4096, 4096, 4096, 4096)
cases["base +128"] = (128, 0, 0, 0)
cases["base +1 MB"] = (1 << 20, 0, 0, 0)
outs = {k: carve(v) for k, v in cases.items()}
times = {k: [] for k in cases}
for r in range(4):
    for k in cases:
        ms = timed(outs[k], 10)
        if r:
            times[k].append(ms)
total = sum(sizes)
print("%s B=%d no=%d nc=%d: %.2f GB per launch; P at %x" % (which, B, no, nc, total / 1e9, outs["packed"][0].data_ptr()))
for k, v in times.items():
    med = float(np.median(v))
    print("  %-16s median %.3f ms (min %.3f max %.3f)  %.3f of 8 TB/s" % (k, med, min(v), max(v), total / med / 1e6 / 8000))

# ---- separate allocations of the same four arrays: does it matter WHICH memory they got?
del outs, pool
torch.cuda.empty_cache()
f = dict(dtype=torch.float64, device="cuda")
sets = []
nsets = int(os.environ.get("MPCASM_NSETS", "6"))
for k in range(nsets):
    sets.append((torch.empty((B, no, no), **f), torch.empty((B, no), **f),
                 torch.empty((B, nc, no), **f), torch.empty((B, nc), **f)))
times = [[] for _ in sets]
for r in range(4):
    for k, out in enumerate(sets):
        ms = timed(out, 10)
        if r:
            times[k].append(ms)
for k, v in enumerate(times):
    med = float(np.median(v))
    print("  allocation %d (P at %x, G at %x)  median %.3f ms (min %.3f max %.3f)  %.3f of 8 TB/s"
          % (k, sets[k][0].data_ptr(), sets[k][2].data_ptr(), med, min(v), max(v), total / med / 1e6 / 8000))

# ---- which of the four arrays decides, and does a plain fill see the same difference?
med = [float(np.median(v)) for v in times]
fast, slow = int(np.argmin(med)), int(np.argmax(med))
print("  fastest set %d (%.3f ms), slowest %d (%.3f ms)" % (fast, med[fast], slow, med[slow]))
if med[slow] > 1.15 * med[fast]:
    names = "PqGh"
    for k in range(4):
        mix = tuple(sets[slow][j] if j == k else sets[fast][j] for j in range(4))
        t = min(timed(mix, 10) for _ in range(3))
        mix2 = tuple(sets[fast][j] if j == k else sets[slow][j] for j in range(4))
        t2 = min(timed(mix2, 10) for _ in range(3))
        print("  only %s from the slow set: %.3f ms;   only %s from the fast set: %.3f ms" % (names[k], t, names[k], t2))
    for label, k in (("fast", fast), ("slow", slow)):
        for j in (0, 2):
            x = sets[k][j]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x.fill_(1.0)
            e0.record()
            for _ in range(5):
                x.fill_(1.0)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("  fill of %s of the %s set: %.3f ms = %.2f TB/s" % (names[j], label, ms, x.numel() * 8 / ms / 1e9))

    # ---- the order of the instances in time (runs of 2^k consecutive instances per workgroup)
    from mpcasm import capi
    lib = capi.load()
    for label, bits in (("runs of 4 (shipped)", 0), ("single instances", 256), ("runs of 2", 1 << 10), ("runs of 8", 3 << 10),
                        ("runs of 16", 4 << 10), ("runs of 64", 6 << 10), ("XCD-wise, runs of 4", 8192),
                        ("XCD-wise, runs of 16", 8192 | (4 << 10)), ("XCD-wise, single", 8192 | 256),
                        ("streams not skewed", 16384), ("shipped again", 0)):
        lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT | bits)
        timed(sets[fast], 3)
        tf = min(timed(sets[fast], 10) for _ in range(3))
        ts = min(timed(sets[slow], 10) for _ in range(3))
        print("  %-20s fast set %.3f ms, slow set %.3f ms" % (label, tf, ts))
    lib.mpcasm_set_option(capi.OPT_PHASE_MASK, capi.PHASE_DEFAULT)
