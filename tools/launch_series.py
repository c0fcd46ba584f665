"""Durations of successive launches of the C2 assembly right after start-up (clocks settle over
some tens of milliseconds of load): python tools/launch_series.py [batch [launches [group]]]."""
import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd")); sys.path.insert(0, ROOT)
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 40
GRP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
work = bench.build_workload(4096, 1)
tile = lambda x: np.concatenate([x] * ((B + 4095) // 4096))[:B]
asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
asm.bind_lti("LIP", torch.as_tensor(tile(work["A"]), device="cuda"), torch.as_tensor(tile(work["B"]), device="cuda"))
given = torch.as_tensor(tile(work["given"]), device="cuda")
outs = [tuple(torch.empty_like(t) for t in asm.assemble(given)) for _ in range(3)]
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NL // GRP + 1)]
ev[0].record()
for k in range(NL):
    asm.assemble(given, out=outs[k % 3])
    if (k + 1) % GRP == 0:
        ev[(k + 1) // GRP].record()
torch.cuda.synchronize()
print("us per launch, groups of %d:" % GRP, " ".join("%.1f" % (ev[k].elapsed_time(ev[k + 1]) * 1e3 / GRP) for k in range(NL // GRP)))
