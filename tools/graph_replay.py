#!/usr/bin/env python3
"""The C2 bench step replayed from a hipGraph (three captured launches, one per output set) against
the same launches issued one by one: what the launch gaps are worth.  python tools/graph_replay.py [batch]"""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    work = bench.build_workload(B, 1)
    asm = work["engine"].Assembler(work["form"], batch=B, lti=["LIP"])
    asm.bind_lti("LIP", torch.as_tensor(work["A"], device="cuda"), torch.as_tensor(work["B"], device="cuda"))
    given = torch.as_tensor(work["given"], device="cuda")
    outs = [tuple(torch.empty_like(t) for t in asm.assemble(given)) for _ in range(3)]
    torch.cuda.synchronize()

    def three():
        for o in outs:
            asm.assemble(given, out=o)

    plain = bench._event_ms(torch, three, 300) / 3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        three()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        three()
    replay = bench._event_ms(torch, graph.replay, 300) / 3
    ref = [tuple(t.clone() for t in o) for o in outs]
    for o in outs:
        for t in o:
            t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for o, r in zip(outs, ref) for a, b in zip(o, r))
    print("B=%d: %.2f us per launch issued one by one, %.2f us replayed from a graph of three; results identical: %s"
          % (B, plain * 1e3, replay * 1e3, same))


if __name__ == "__main__":
    main()
