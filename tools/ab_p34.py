import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/mpc-interface_amd")
import numpy as np, torch
import bench
from mpcasm import capi
lib = capi.load()
dev = torch.device("cuda", 0)
for opt in (0, 1, 2):
    lib.mpcasm_set_option(capi.OPT_P_DIRECT if hasattr(capi, "OPT_P_DIRECT") else 5, opt)
    recs = bench.variant_records(torch, dev, 4096, 20260)
    print("P_DIRECT", opt, [(r["no"], r["nc"], "%.1f us" % (r["avg_launch_ms"] * 1e3)) for r in recs])
