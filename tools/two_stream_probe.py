"""Probe: how much do two INDEPENDENT half-batch training passes gain from running concurrently on two streams -- the
matrix-bound kernels of one beside the memory-bound kernels of the other?  Two model replicas, forward + backward of B
samples each captured into a hipGraph on its own stream (no host time in the way); N replays of both graphs one after the
other on one stream vs concurrently on two.  Run under different TDX_PERSISTENT_CUS (the persistent conv kernels fill a CU's
registers and LDS: with 256 of them nothing else can run beside).  GPU box:  python tools/two_stream_probe.py [B] [N]"""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
from types import SimpleNamespace
import bench
from turbdiff_amd.models.conditioning import Conditioning

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def make():
    diff = bench.build_model(dev, torch.bfloat16)
    x, c, idx = bench.synthetic_inputs(B, dev)
    C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
    params = [p for p in diff.parameters() if p.requires_grad]

    def step():
        for p in params: p.grad = None
        loss, _ = diff(x, C, md, None)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    for p in params: p.grad = None
    with torch.cuda.graph(g, stream=s):
        loss = step()
    return g, s, loss, (diff, x, C, md)


ga, sa, la, keep_a = make()
gb, sb, lb, keep_b = make()
for _ in range(3):
    ga.replay(); gb.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    ga.replay(); gb.replay()
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / N * 1e3
t0 = time.perf_counter()
for _ in range(N):
    with torch.cuda.stream(sa): ga.replay()
    with torch.cuda.stream(sb): gb.replay()
torch.cuda.synchronize()
con = (time.perf_counter() - t0) / N * 1e3
import os
print(f"TDX_PERSISTENT_CUS={os.environ.get('TDX_PERSISTENT_CUS', '256')} B={B}x2: fwd+bwd of both replicas {seq:.2f} ms one after the other, "
      f"{con:.2f} ms on two streams ({seq / con:.3f}x); losses {la.item():.4f} {lb.item():.4f}")
