import os, sys, time, cProfile, pstats
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd")); sys.path.insert(0, ROOT)
import torch
from mpcasm import problems
from mpcasm.walkers import WalkerFleet
B=4096
fleet = WalkerFleet(B, conf=problems.BipedConfig(step_samples=8), graphs="graphs" in sys.argv)
g = torch.zeros((B, fleet.given_len), dtype=torch.float64, device="cuda")
for _ in range(40): fleet.tick(g)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(64): fleet.tick(g)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host per tick %.1f us, incl. sync %.1f us"%((t1-t)/64*1e6,(t2-t)/64*1e6))
pr=cProfile.Profile(); pr.enable()
for _ in range(64): fleet.tick(g)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
