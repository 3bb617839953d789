#!/usr/bin/env python3
"""Which torch ops of one eager forward + backward issue device-to-device MEMCPYs (they become memcpy nodes in a captured step)?
torch.profiler with stacks; prints every 'Memcpy DtoD' with the aten op and the Python frames that led to it."""
import sys
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from turbdiff_amd.models.conditioning import Conditioning

dev = torch.device("cuda:0")
diff = bench.build_model(dev); bench.set_mode(diff, "bf16")
x, c, idx = bench.synthetic_inputs(2, dev, (96, 32, 24))
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
for _ in range(2):
    diff.zero_grad(set_to_none=True)
    loss, _ = diff(x, C, md, None); loss.backward()
torch.cuda.synchronize()
diff.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    loss, _ = diff(x, C, md, None)
    loss.backward()
    torch.cuda.synchronize()
ev = prof.events()
rt = [e for e in ev if e.name == "hipMemcpyAsync"]
print(len(rt), "hipMemcpyAsync calls in one forward + backward")
for m in rt:
    t = m.time_range.start
    owners = sorted((e for e in ev if e is not m and e.time_range.start <= t <= e.time_range.end), key=lambda e: e.time_range.elapsed_us())
    names = [o.name for o in owners if not o.name.startswith("hip")][:8]
    shapes = [str(getattr(o, "input_shapes", "")) for o in owners if o.name in ("aten::cat", "aten::copy_")][:1]
    print("--", " <- ".join(names), shapes)
