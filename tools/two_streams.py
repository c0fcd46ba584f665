#!/usr/bin/env python3
"""The C2 bench step issued on one stream against the same steps dealt to two streams in turn
(independent steps: a launch may start on the CUs the previous one has left).
python tools/two_streams.py [batch [steps]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    work = bench.build_workload(B, 1)
    from mpcasm import capi
    capi.load().mpcasm_set_option(capi.OPT_P_DIRECT, int(os.environ.get("MPCASM_P_DIRECT", "0")))
    asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
    capi.load().mpcasm_set_option(capi.OPT_P_DIRECT, 0)
    asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
    given = torch.as_tensor(work["given"], device="cuda")
    outs = [tuple(torch.empty_like(t) for t in asm.assemble(given)) for _ in range(8)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]

    def run(nstreams):
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < 40.0:          # the clocks settle
            for k in range(16):
                asm.assemble(given, out=outs[k % 8], stream=streams[k % nstreams])
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for s in streams[1:nstreams]:
            s.wait_stream(streams[0])
        e0.record(streams[0])
        for s in streams[1:nstreams]:
            s.wait_event(e0)
        for k in range(K):
            asm.assemble(given, out=outs[k % 8], stream=streams[k % nstreams])
        for s in streams[1:nstreams]:
            streams[0].wait_stream(s)
        e1.record(streams[0])
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K * 1e3

    for n in (1, 2, 1, 2):
        print("B=%d, %d stream(s): %.2f us per step" % (B, n, run(n)))


if __name__ == "__main__":
    main()
