"""time tdx_encode_fwd: the 16 raw channels of the composed first conv and the 64-channel encoder output"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd import ops
d = torch.device("cuda")
grid = (192, 64, 48)
for B in (6, 8):
    x = torch.randn(B, 4, *grid, device=d)
    c = torch.randn(4, *grid, device=d)
    for D in (8, 32):
        wx, bx = torch.randn(D, 4, 1, 1, 1, device=d), torch.randn(D, device=d)
        wc, bc = torch.randn(D, 4, 1, 1, 1, device=d), torch.randn(D, device=d)
        for _ in range(3):
            y = ops.encode(x, c, wx, bx, wc, bc, torch.bfloat16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.encode(x, c, wx, bx, wc, bc, torch.bfloat16)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (x.numel() * 4 + c.numel() * 4 + y.numel() * 2) / 1e6
        print(f"B {B} channels {2 * D:3d}: {us:7.1f} us  {mb / us * 1e-3 * 1e3 / 1e3:.2f} TB/s  checksum {float(y.float().sum()):.3f}")
