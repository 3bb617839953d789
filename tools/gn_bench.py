#!/usr/bin/env python3
"""GroupNorm backward (tdx_gn_bwd: reduce pass + group pass + apply pass) at the U-Net's level-0 / level-1 shapes, B = 6, bf16:
microseconds per call and the bandwidth over its 5 activation passes.  (Round 3 measured an apply pass that walks each
sample back to front, so that what the reduce pass left in the 256 MiB Infinity Cache is read first: 199 vs 192 us at
192x64x48 x 32 channels, 437 vs 431 at 64 channels -- no gain, not kept.)
GPU box: python tools/gn_bench.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "generative-turbulence_amd"))
import torch
from turbdiff_amd import _lib as L

dev = torch.device("cuda:0")
B = 6
for (grid, C) in (((192, 64, 48), 64), ((192, 64, 48), 32), ((96, 32, 24), 128), ((96, 32, 24), 64)):
    V = grid[0] * grid[1] * grid[2]
    x = torch.randn(B, V, C, device=dev).bfloat16(); dy = torch.randn(B, V, C, device=dev).bfloat16(); dx = torch.empty_like(x)
    f = lambda *s: torch.randn(*s, device=dev)
    gamma, beta, scale, shift = f(C), f(C), 0.1 * f(B, C), f(B, C)
    dg, db, ds, dsh = f(C), f(C), f(B, C), f(B, C)
    G = 8
    ws = torch.zeros(L.query("tdx_gn_workspace_bytes", B, C) + (1 << 24), dtype=torch.uint8, device=dev)
    stats = torch.empty(B, G, 2, device=dev)
    st = L.stream()
    L.call("tdx_gn_stats", L.ptr(x), L.ptr(stats), B, V, C, G, 1e-5, L.BF16, L.ptr(ws), st)
    out = []
    for rep in range(2):
        go = lambda: L.call("tdx_gn_bwd", L.ptr(x), L.ptr(dy), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift),
                            L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ds), L.ptr(dsh), B, V, C, G, 1, L.BF16, L.ptr(ws), st)
        for _ in range(5): go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append(f"{us:7.1f} us ({5 * x.numel() * 2 / us / 1e6:5.2f} TB/s)")
    print(f"gn_bwd {grid[0]}x{grid[1]}x{grid[2]} C={C:3d} ({x.numel() * 2 / 1e6:.0f} MB per tensor): " + "  ".join(out), flush=True)
