"""time tdx_resize_fwd on the U-Net's up-sampling shapes (TDX_RESIZE_PAIRS=0: one output per lane)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd import ops
d = torch.device("cuda")
shapes = [((96, 32, 24), (192, 64, 48), 64), ((48, 16, 12), (96, 32, 24), 128), ((24, 8, 6), (48, 16, 12), 256),
          ((97, 25, 25), (194, 50, 50), 64)]
for B in (6, 8):
    for si, so, C in shapes:
        x = torch.randn(B, *si, C, device=d).to(torch.bfloat16)
        for _ in range(3):
            y = ops.resize(x, list(so))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.resize(x, list(so))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(f"B {B} {si} -> {so} C {C:3d}: {us:7.1f} us  {mb / us / 1e3 * 1e3 / 1e3:.2f} TB/s  sum {float(y.float().abs().mean()):.5f}")
