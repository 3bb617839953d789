#!/usr/bin/env python3
"""Timing of the regression baselines' conv layers at the shapes the reference's models use them at, forward
and forward + backward: the matrix-core kernels (tdx_convg_mfma.hip, the default for bf16 tensors), the vector-ALU kernels
(tdx_convg.hip, TDX_CONVG_MFMA=0) and the same layers through stock PyTorch-ROCm (MIOpen) on the same GPU.
Usage (GPU box): python tools/baseline_conv_bench.py"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import torch.nn.functional as F
from turbdiff_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def case(name, B, Ci, Co, grid, k, stride, dil, pad, mode, transposed=False):
    X, Y, Z = grid
    x = torch.randn(B, X, Y, Z, Ci, device=dev).bfloat16().requires_grad_()
    w = (torch.randn(*( (Ci, Co) if transposed else (Co, Ci) ), k, k, k, device=dev) * 0.05).requires_grad_()
    b = torch.zeros(Co, device=dev, requires_grad=True)
    if transposed:
        f = lambda: ops.conv_transpose3d(x, w, b, stride=stride, padding=pad)
    else:
        f = lambda: ops.conv3d(x, w, b, stride=stride, dilation=dil, padding=pad, padding_mode=mode)
    y = f(); gy = torch.randn_like(y)
    def fb():
        x.grad = w.grad = b.grad = None
        f().backward(gy)
    res = {}
    for sw in ("1", "0"):  # the library reads the switch per call
        os.environ["TDX_CONVG_MFMA"] = sw
        with torch.no_grad():
            res[sw] = (timeit(f), None)
        res[sw] = (res[sw][0], timeit(fb, 3))
    os.environ["TDX_CONVG_MFMA"] = "1"
    (tf, tfb), (vf, vfb) = res["1"], res["0"]
    # stock path: NCDHW bf16
    xs = x.detach().permute(0, 4, 1, 2, 3).contiguous().requires_grad_()
    ws = w.detach().bfloat16().requires_grad_(); bs = b.detach().bfloat16().requires_grad_()
    if transposed:
        g = lambda: F.conv_transpose3d(xs, ws, bs, stride=stride, padding=pad)
    elif mode == "replicate":
        g = lambda: F.conv3d(F.pad(xs, (pad,) * 6, mode="replicate"), ws, bs, stride=stride, dilation=dil)
    else:
        g = lambda: F.conv3d(xs, ws, bs, stride=stride, dilation=dil, padding=pad)
    try:
        with torch.no_grad():
            sf = timeit(g, 3)
        ys = g(); gys = torch.randn_like(ys)
        def gb():
            xs.grad = ws.grad = bs.grad = None
            g().backward(gys)
        sfb = timeit(gb, 2)
    except Exception as e:  # noqa: BLE001
        sf = sfb = float("nan")
    vo = y.numel() // Co
    fl = 2.0 * k ** 3 * Ci * Co * (vo if not transposed else x.numel() // Ci) / 1e9
    print(f"{name:28s} {B}x{X}x{Y}x{Z} {Ci:3d}->{Co:3d} k{k} s{stride} d{dil} | matrix-core fwd {tf:6.3f} ms ({fl/tf:5.0f} TF/s) f+b {tfb:7.3f} | "
          f"vector-ALU fwd {vf:6.3f} f+b {vfb:7.3f} | PyTorch-ROCm fwd {sf:6.3f} f+b {sfb:8.3f} ms", flush=True)


print("baseline conv variants, bf16 NDHWC: tdx_convg_mfma.hip (matrix cores), tdx_convg.hip (vector ALU), stock PyTorch-ROCm (bf16 NCDHW, MIOpen)")
for d in (1, 2, 4, 8):  # DilatedCNNBlock, dilresnet.py:22-44: 48 channels at the data resolution
    case(f"dilresnet conv dilation {d}", 2, 48, 48, (96, 64, 48), 3, 1, d, d, "replicate")
case("tfnet conv k3 stride 2", 2, 64, 128, (96, 64, 48), 3, 2, 1, 1, "zeros")   # tfnet.py:185-199
case("tfnet conv k3 stride 1", 2, 64, 64, (96, 64, 48), 3, 1, 1, 1, "zeros")
case("tfnet deconv k4 stride 2", 2, 128, 64, (48, 32, 24), 4, 2, 1, 1, "zeros", transposed=True)  # tfnet.py:201-208
