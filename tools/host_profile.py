#!/usr/bin/env python3
"""cProfile of the host side of the training step (where does the Python time go?)."""
import cProfile, pstats, sys, time
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning
dev = torch.device("cuda:0")
diff = bench.build_model(dev, torch.bfloat16)
from turbdiff_amd.optim import ClipRAdam
opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1)
x, c, idx = bench.synthetic_inputs(6, dev)
C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
def step():
    loss, _ = diff(x, C, md, None); loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
# pure host time: enqueue only (no sync inside), then sync
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/5:.2f} ms/step, total {1e3*(t2-t0)/5:.2f} ms/step")
def phases():
    t = time.perf_counter(); loss, _ = diff(x, C, md, None); a = time.perf_counter()
    loss.backward(); b = time.perf_counter()
    c_ = time.perf_counter()
    opt.step(); d = time.perf_counter(); opt.zero_grad(set_to_none=True); e = time.perf_counter()
    return [1e3*(v) for v in (a-t, b-a, c_-b, d-c_, e-d)]
torch.cuda.synchronize()
acc = [0]*5
for _ in range(5):
    p = phases(); acc = [u+v for u, v in zip(acc, p)]
torch.cuda.synchronize()
print("host ms/step: fwd %.2f  bwd %.2f  clip %.2f  opt %.2f  zero %.2f" % tuple(v/5 for v in acc))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
