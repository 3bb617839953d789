#!/usr/bin/env python3
"""BASELINE config 5: Attention(dim=128, heads=4, dim_head=32) core at 96x32x24 (N = 73 728 tokens),
MFMA flash kernels with fp16 (the config's wording) or bf16 operands (--dtype).  Reports time, algorithmic TFLOP/s
(4 N^2 d h) and Q/K/V/O GB/s."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "generative-turbulence_amd"))
import torch
from turbdiff_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--n", type=int, default=96 * 32 * 24)
ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
a = ap.parse_args()
B, H, D, N = a.batch, 4, 32, a.n
dev = torch.device("cuda:0")
qkv = torch.randn(B, N, 3 * H * D, device=dev).to(torch.float16 if a.dtype == "f16" else torch.bfloat16)
for _ in range(2): ops.attention(qkv, H)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
s.record()
for _ in range(n): ops.attention(qkv, H)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
flops = 4.0 * N * N * D * H * B
byts = 4.0 * N * H * D * 2 * B
qg = qkv.clone().requires_grad_()
out = ops.attention(qg, H)
go = torch.randn_like(out)
for _ in range(2): out.backward(go, retain_graph=True)
torch.cuda.synchronize()
s.record()
for _ in range(n): out.backward(go, retain_graph=True)
e.record(); torch.cuda.synchronize()
msb = s.elapsed_time(e) / n
print(f"attention ({a.dtype} operands) bwd N={N} B={B}: {msb:.3f} ms  {3.5 * flops/msb/1e9:.0f} TFLOP/s executed (14 N^2 d h: dQ pass 6, dK/dV pass 8)")
print(f"attention ({a.dtype} operands) fwd N={N} B={B}: {ms:.3f} ms  {flops/ms/1e9:.0f} TFLOP/s ({flops/ms/1e9/2500*100:.1f}% of 2.5 PF)  "
      f"algorithmic Q/K/V/O traffic {byts/ms/1e6:.1f} GB/s ({byts/ms/1e6/8000*100:.2f}% of 8 TB/s)")
