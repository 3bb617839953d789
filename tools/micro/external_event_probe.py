#!/usr/bin/env python3
"""Does an event recorded INSIDE a captured hipGraph (an event-record node: hipEventRecordWithFlags(...,
hipEventRecordExternal) on the capturing stream) release a graph-external stream while the rest of the graph is still
running?  That is what would let the gradient all-reduce of a bucket start behind the captured backward's bucket boundary
(parallel.BucketedDataParallel with training.GraphedTrainingStep).  torch.cuda.Event(external=True) refuses on ROCm
("External events are disallowed in rocm"), so the probe talks to the HIP runtime directly.
Prints the timeline of three replays; exit code 0 = works, 3 = the side stream only ran after the whole graph, 4 = the
runtime refused."""
import ctypes
import sys
import torch

hip = ctypes.CDLL("libamdhip64.so")
d = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=d)
out = torch.empty_like(a)
side_buf = torch.zeros(1 << 20, device=d)
s, side = torch.cuda.Stream(), torch.cuda.Stream()
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0  # hipEventDisableTiming
g = torch.cuda.CUDAGraph()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        torch.mm(a, a, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        torch.mm(a, a, out=out)          # "backward up to the bucket boundary"
        rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(s.cuda_stream), 1)  # hipEventRecordExternal: an event-record node
        for _ in range(6):
            torch.mm(a, a, out=out)      # "the rest of backward"
print("hipEventRecordWithFlags(external) during capture ->", rc)
if rc != 0:
    sys.exit(4)
torch.cuda.synchronize()
t0, t_side, t_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for rep in range(3):
    t0.record()
    g.replay()
    rc = hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0)
    with torch.cuda.stream(side):
        side_buf.add_(1.0)
        t_side.record()
    t_end.record()
    torch.cuda.synchronize()
    a_ms, e_ms = t0.elapsed_time(t_side), t0.elapsed_time(t_end)
    print(f"replay {rep}: wait rc {rc}; side stream released at {a_ms:.2f} ms, graph done at {e_ms:.2f} ms")
ok = a_ms < 0.6 * e_ms
print("external event inside a captured graph:", "releases the side stream early" if ok else "NOT early")
sys.exit(0 if ok else 3)
