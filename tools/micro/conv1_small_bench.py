"""time tdx_conv1_fwd on the launches that cannot fill the chip (deep levels, B = 1): rows x K x N sweep"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd import _lib as L
d = torch.device("cuda")
for rows, K, N in [(1152, 128, 256), (1152, 256, 256), (1152, 512, 256), (1152, 1024, 256), (144, 512, 384), (144, 512, 512),
                   (9216, 512, 128), (9216, 128, 128), (73728, 256, 64), (73728, 64, 64)]:
    x = torch.randn(rows, K, device=d).to(torch.bfloat16)
    w = torch.randn(K, N, device=d)
    b = torch.randn(N, device=d)
    y = torch.empty(rows, N, device=d, dtype=torch.bfloat16)
    go = lambda: L.call("tdx_conv1_fwd", L.ptr(x), K, None, 0, L.ptr(w), N, L.ptr(b), None, L.ptr(y), rows, N, L.BF16, L.stream())
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    wgs = -(-rows // 256) * (N // 64)
    print(f"rows {rows:6d} K {K:5d} N {N:4d}: {us:6.1f} us   {wgs:4d} workgroups x {K // 32:3d} slices -> {us / (K // 32):5.2f} us per slice")
