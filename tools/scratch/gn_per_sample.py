"""Experiment: GroupNorm backward called once for the whole batch (reduce pass over 6 samples, then apply pass over 6 samples:
900 MB between the two reads of a line at level 0) against sample by sample (reduce + apply of ONE sample back to back: 150 MB
between the reads, inside the 256 MiB Infinity Cache)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "generative-turbulence_amd"))
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

    def whole():
        L.call("tdx_gn_bwd", L.ptr(x), L.ptr(dy), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift),
               L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ds), L.ptr(dsh), B, V, C, G, 1, L.BF16, L.ptr(ws), st)

    def per(n):
        for b in range(0, B, n):
            L.call("tdx_gn_bwd", L.ptr(x[b:b + n]), L.ptr(dy[b:b + n]), L.ptr(stats[b:b + n]), L.ptr(gamma), L.ptr(beta), L.ptr(scale[b:b + n]),
                   L.ptr(shift[b:b + n]), L.ptr(dx[b:b + n]), L.ptr(dg), L.ptr(db), L.ptr(ds[b:b + n]), L.ptr(dsh[b:b + n]), n, V, C, G, 1,
                   L.BF16, L.ptr(ws), st)

    def timeit(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3

    res = [("whole batch", timeit(whole))] + [(f"{n} sample(s) per call", timeit(lambda n=n: per(n))) for n in (1, 2, 3)]
    print(f"gn_bwd {grid[0]}x{grid[1]}x{grid[2]} C={C:3d} ({x.numel() * 2 / B / 1e6:.0f} MB per sample and tensor): "
          + "  ".join(f"{k}: {v:6.1f} us" for k, v in res), flush=True)
