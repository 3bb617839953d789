"""time tdx_resize_bwd (adjoint of the up path's trilinear up-sampling) on the U-Net's shapes; TDX_RESIZE_BWD_TILES=0: eight-pass tiles"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd import _lib as L
d = torch.device("cuda")
B = 6
for si, so, C in [((96, 32, 24), (192, 64, 48), 64), ((48, 16, 12), (96, 32, 24), 128), ((24, 8, 6), (48, 16, 12), 256), ((12, 4, 3), (24, 8, 6), 512)]:
    dy = torch.randn(B, *so, C, device=d).to(torch.bfloat16)
    dx = torch.empty(B, *si, C, device=d, dtype=torch.bfloat16)
    go = lambda: L.call("tdx_resize_bwd", L.ptr(dy), None, L.ptr(dx), B, *si, *so, C, L.BF16, L.stream())
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    print(f"{si} <- {so} C {C:3d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   checksum {float(dx.float().abs().mean()):.5f}")
