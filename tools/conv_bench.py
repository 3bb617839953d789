#!/usr/bin/env python3
"""Per-layer timing of the 3x3x3 conv kernels at the BASELINE config-2 shapes (B per GPU = 6).
Usage (GPU box): python tools/conv_bench.py [--batch 6] [--impl auto]"""
import argparse, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
from turbdiff_amd import _lib as L, ops

LAYERS = [  # name, C1, C2, Cout, grid
    ("down.0.b1", 64, 0, 64, (192, 64, 48)), ("down.1.b1", 64, 0, 128, (96, 32, 24)), ("down.1.b2", 128, 0, 128, (96, 32, 24)),
    ("down.2.b1", 128, 0, 256, (48, 16, 12)), ("down.2.b2", 256, 0, 256, (48, 16, 12)), ("down.3.b1", 256, 0, 512, (24, 8, 6)),
    ("down.3.b2", 512, 0, 512, (24, 8, 6)), ("center", 512, 0, 512, (12, 4, 3)), ("up.0.b1", 512, 512, 256, (24, 8, 6)),
    ("up.0.b2", 256, 0, 256, (24, 8, 6)), ("up.1.b1", 256, 256, 128, (48, 16, 12)), ("up.1.b2", 128, 0, 128, (48, 16, 12)),
    ("up.2.b1", 128, 128, 64, (96, 32, 24)), ("up.2.b2", 64, 0, 64, (96, 32, 24)), ("up.3.b1", 64, 64, 32, (192, 64, 48)),
    ("up.3.b2", 32, 0, 32, (192, 64, 48)),
]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def miopen_layer(B, Ci, Co, grid, dtype, channels_last):
    """The same layer through stock PyTorch-ROCm: F.pad(replicate) + F.conv3d (MIOpen), NCDHW or
    channels_last_3d.  Returns (fwd ms, fwd+bwd ms)."""
    import torch.nn.functional as F
    X, Y, Z = grid
    x = torch.randn(B, Ci, X, Y, Z, device="cuda", dtype=dtype)
    w = (torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.02).to(dtype)
    b = torch.zeros(Co, device="cuda", dtype=dtype)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last_3d); w = w.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(); w.requires_grad_(); b.requires_grad_()
    fwd = lambda: F.conv3d(F.pad(x, (1,) * 6, mode="replicate"), w, b)
    with torch.no_grad():
        tf = timeit(fwd, 3)
    gy = torch.randn_like(fwd())
    def both():
        x.grad = w.grad = b.grad = None
        fwd().backward(gy)
    return tf, timeit(both, 3)

def miopen_table(a):
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    print(f"stock PyTorch-ROCm conv3d (MIOpen, find={'on' if a.miopen_find else 'off'}), B={a.batch}; TF/s on 54*Cin*Cout*V (fwd) and 3x that (fwd+bwd)")
    print(f"{'layer':12s} {'Cin':>5s} {'Cout':>5s} | " + " | ".join(f"{n:>24s}" for n in ("bf16 NCDHW fwd / f+b ms", "bf16 NDHWC fwd / f+b ms", "fp32 NCDHW fwd / f+b ms")))
    tot = [[0.0, 0.0] for _ in range(3)]; totf = 0.0
    for name, C1, C2, Co, grid in LAYERS:
        if a.only and a.only not in name: continue
        Ci = C1 + C2; fl = 54.0 * Ci * Co * a.batch * grid[0] * grid[1] * grid[2]
        mult = 4 if name == "center" else (2 if name == "down.0.b1" else (3 if name == "up.3.b2" else 1))
        cells = []
        for k, (dt, cl) in enumerate([(torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, False)]):
            try:
                tf, tb = miopen_layer(a.batch, Ci, Co, grid, dt, cl)
            except Exception as e:  # noqa: BLE001 -- report and continue with the other layouts
                cells.append(f"{'failed: ' + type(e).__name__:>24s}"); tot[k][0] = tot[k][1] = float("nan"); continue
            tot[k][0] += tf * mult; tot[k][1] += tb * mult
            cells.append(f"{tf:7.2f} {fl/tf/1e9:4.0f} /{tb:7.2f} {3*fl/tb/1e9:4.0f}")
        totf += fl * mult
        print(f"{name:12s} {Ci:5d} {Co:5d} | " + " | ".join(cells) + f"   x{mult}", flush=True)
    for k, n in enumerate(("bf16 NCDHW", "bf16 NDHWC", "fp32 NCDHW")):
        print(f"TOTAL {n}: fwd {tot[k][0]:.1f} ms ({totf/tot[k][0]/1e9:.0f} TF/s)  fwd+bwd {tot[k][1]:.1f} ms ({3*totf/tot[k][1]/1e9:.0f} TF/s)")

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=6); ap.add_argument("--only", default="")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--no-wgrad", action="store_true", help="skip the weight-gradient column")
    ap.add_argument("--layers", default="", help="comma-separated layer names (exact), e.g. down.0.b1,up.3.b2")
    ap.add_argument("--zeros", action="store_true", help="all-zero activations and weights (switching-activity experiment)")
    ap.add_argument("--miopen", action="store_true", help="time the same layers through F.conv3d (MIOpen) instead")
    ap.add_argument("--miopen-find", action="store_true", help="with --miopen: torch.backends.cudnn.benchmark = True")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp16"])
    ap.add_argument("--bf16-values", action="store_true",
                    help="fp16 tensors whose VALUES are bf16-representable (3 low mantissa bits zero): same kernels and opcodes as "
                         "--dtype fp16, the operand bits of --dtype bf16 -- separates the opcode from the switching activity")
    ap.add_argument("--impl", default="auto", choices=["auto", "direct", "mfma", "split"])
    a = ap.parse_args()
    if a.miopen:
        return miopen_table(a)
    dev = torch.device("cuda:0"); B = a.batch
    tdt, dc = {"bf16": (torch.bfloat16, 1), "fp16": (torch.float16, 3), "f32": (torch.float32, 0)}[a.dtype]
    coarse = (lambda t: t.to(torch.bfloat16).to(tdt)) if a.bf16_values else (lambda t: t.to(tdt))
    os.environ["TDX_CONV_IMPL"] = a.impl; im = L.conv_impl()
    tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}; totf = 0
    print(f"{'layer':12s} {'Cin':>5s} {'Cout':>5s} {'grid':>12s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
    for name, C1, C2, Co, (X, Y, Z) in LAYERS:
        if a.only and a.only not in name: continue
        if a.layers and name not in a.layers.split(","): continue
        Ci = C1 + C2
        zf = 0.0 if a.zeros else 1.0
        x1 = coarse(torch.randn(B, X, Y, Z, C1, device=dev) * zf)
        x2 = coarse(torch.randn(B, X, Y, Z, C2, device=dev)) if C2 else None
        w = (torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02 * zf)
        if a.bf16_values:
            w = w.to(torch.bfloat16).float()
        bias = torch.zeros(Co, device=dev)
        gy = coarse(torch.randn(B, X, Y, Z, Co, device=dev) * zf)
        wf, wb = ops._packed_conv3(w, tdt)
        y = torch.empty(B, X, Y, Z, Co, device=dev, dtype=tdt)
        st = L.stream()
        f = lambda: L.call("tdx_conv3_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), B, X, Y, Z, Co, dc, im, st)
        gx1 = torch.empty_like(x1); gx2 = torch.empty_like(x2) if C2 else None
        ws = torch.empty(L.query("tdx_conv3_bwd_data_workspace_bytes", B, X, Y, Z, Ci, dc, 0), dtype=torch.uint8, device=dev)
        d = lambda: L.call("tdx_conv3_bwd_data", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, 0, B, X, Y, Z, Co, dc, im, L.ptr(ws), st)
        gw = torch.empty_like(w); gb = torch.empty(Co, device=dev)
        ws2 = torch.empty(L.query("tdx_conv3_bwd_weight_workspace_bytes", Ci, Co, 0), dtype=torch.uint8, device=dev)
        wg = lambda: L.call("tdx_conv3_bwd_weight", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(gy), L.ptr(gw), L.ptr(gb), B, X, Y, Z, Co, dc, im, L.ptr(ws2), st)
        fl = 54.0 * Ci * Co * B * X * Y * Z
        tf, td, tw = (timeit(f, 10), 1e9, 1e9) if a.fwd_only else (timeit(f), timeit(d), 1e9 if a.no_wgrad else timeit(wg))
        mult = 4 if name == "center" else (2 if name in ("down.0.b1",) else 1)  # center x4; down.0.b1 == down.0.b2
        if name == "up.3.b2": mult = 3  # + decode.0.b1, decode.0.b2
        for k, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw)): tot[k] += t * mult
        totf += fl * mult
        print(f"{name:12s} {Ci:5d} {Co:5d} {X:4d}x{Y:3d}x{Z:3d} | {tf:8.3f} {fl/tf/1e9:6.0f} | {td:8.3f} {fl/td/1e9:6.0f} | {tw:8.3f} {fl/tw/1e9:6.0f}   x{mult}")
    print(f"TOTAL per step (weighted): fwd {tot['fwd']:.2f} ms ({totf/tot['fwd']/1e9:.0f} TF/s)  dgrad {tot['dgrad']:.2f} ms ({totf/tot['dgrad']/1e9:.0f})  wgrad {tot['wgrad']:.2f} ms ({totf/tot['wgrad']/1e9:.0f})")

if __name__ == "__main__":
    main()
