"""Who launches the stray torch elementwise kernels of a training step?  Runs the benchmark step under torch.profiler
with Python stacks and prints, per aten op that launches a device kernel outside libtdx (fill_, zero_, copy_, add, ...),
the call sites in this repo's code, with counts per step.  `python tools/fill_census.py [mode] [B]`."""
import sys
from collections import Counter
from types import SimpleNamespace

sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.optim import ClipRAdam

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
diff = bench.build_model(dev)
bench.set_mode(diff, mode)
x, c, idx = bench.synthetic_inputs(B, dev)
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
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()

# device kernels by name
kern = Counter()
ktime = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern[ev.name[:90]] += 1
        ktime[ev.name[:90]] += ev.device_time if hasattr(ev, "device_time") else ev.cuda_time
print(f"== device kernels that are not libtdx's (per step, {N} steps profiled)")
for k, n in kern.most_common():
    if "at::native" in k or "rocclr" in k or "Cijk" in k or "Memset" in k.lower() or "memcpy" in k.lower():
        print(f"{n / N:7.1f}  {ktime[k] / N:9.1f} us  {k}")

# aten ops with their innermost repo frame
sites = Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum",
                       "aten::cat", "aten::index_select", "aten::index_add_", "aten::silu", "aten::silu_backward",
                       "aten::addmm", "aten::mm", "aten::normal_", "aten::random_", "aten::sin", "aten::addcmul",
                       "aten::div", "aten::sub", "aten::neg", "aten::mul_"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name in ("aten::zeros", "aten::zero_", "aten::zeros_like", "aten::full",
                                                              "aten::ones"):
        pass
    frame = next((f for f in (ev.stack or []) if "/root/repo" in f or "turbdiff_amd" in f or "bench.py" in f), None)
    if frame is None:
        frame = "(autograd engine / no python frame): parent " + (ev.cpu_parent.name if ev.cpu_parent is not None else "-")
    sites[(ev.name, frame[-110:])] += 1
print("== aten ops by call site (per step)")
for (name, frame), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:7.1f}  {name:22s} {frame}")
