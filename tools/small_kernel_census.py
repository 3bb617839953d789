#!/usr/bin/env python3
"""Which host-side ops launch the short kernels of a training step?  torch.profiler over 2 steps (B = 6, bf16):
kernels shorter than 8 us grouped by the aten / autograd op that launched them.  GPU box: python tools/small_kernel_census.py"""
import sys
from collections import defaultdict
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import bench
from torch.profiler import ProfilerActivity, profile
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.optim import ClipRAdam

dev = torch.device("cuda:0")
diff = bench.build_model(dev)
bench.set_mode(diff, sys.argv[1] if len(sys.argv) > 1 else "bf16")
x, c, idx = bench.synthetic_inputs(6, dev)
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1)


def step():
    loss, _ = diff(x, C, md, None)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
STACKS = len(sys.argv) > 2 and sys.argv[2] == 'stacks'
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=STACKS, record_shapes=STACKS) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
ev = prof.events()
# kernel -> launching cpu op via correlation: use key_averages grouped by op name with device time
rows = defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA or not e.kernels:
        continue
    for k in e.kernels:
        if k.duration < 8.0:
            rows[(e.name, k.name[:70])][0] += 1
            rows[(e.name, k.name[:70])][1] += k.duration
tot = sum(v[1] for v in rows.values()) / 2
print(f"kernels < 8 us: {sum(v[0] for v in rows.values()) / 2:.0f} launches, {tot:.0f} us per step")
for (op, kern), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n / 2:6.1f} x {t / n:5.1f} us = {t / 2:7.1f} us/step   {op[:38]:38s} {kern}")

if STACKS:  # where do the aten::copy_ / fill_ / add launches come from?  (python tools/small_kernel_census.py bf16 stacks)
    sites = defaultdict(int)
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA or not e.kernels or not e.name.startswith("aten::"):
            continue
        if all(k.duration >= 8.0 for k in e.kernels):
            continue
        frames = [f for f in (e.stack or []) if "turbdiff_amd" in f or "bench.py" in f or "autograd" in f][:3]
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        sites[(e.name, str(e.input_shapes)[:60] + " " + " <- ".join(chain) + " " + " <- ".join(f.split("/")[-1][:60] for f in frames))] += 1
    for (op, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{n / 2:6.1f}  {op:18s} {where}")
