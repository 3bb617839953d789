#!/usr/bin/env python3
"""Does launching a hipGraphExec that is still RUNNING block the host until the earlier launch has finished?  (It would explain
why the captured training step is 0.5 ms slower than the eager one at B = 6: the ~1.5 ms of host work inside hipGraphLaunch of
~700 nodes could not be hidden behind the previous replay.)  A graph of 400 matmul nodes (~10 ms); host time of each of 6
back-to-back replays of ONE graph, then of two graphs captured from the same code replayed alternately."""
import time

import torch

d = torch.device("cuda:0")
a = torch.randn(1024, 1024, device=d, dtype=torch.bfloat16)
o = torch.empty_like(a)
s = torch.cuda.Stream()


def body():
    for _ in range(400):
        torch.mm(a, a, out=o)


def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        body()
    return g


with torch.cuda.stream(s):
    body()
    torch.cuda.synchronize()
    g1, g2 = capture(), capture()
    for name, seq in (("one graph", [g1] * 6), ("two graphs alternating", [g1, g2] * 3)):
        for g in seq[:2]:
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = []
        for g in seq:
            t = time.perf_counter()
            g.replay()
            host.append(1e3 * (time.perf_counter() - t))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:24s}: host ms per replay() call {[round(h, 2) for h in host]}; all issued after {1e3 * (t1 - t0):.2f} ms, "
              f"all done after {1e3 * (t2 - t0):.2f} ms ({1e3 * (t2 - t0) / len(seq):.2f} per replay)")
