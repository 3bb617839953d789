#!/usr/bin/env python3
"""TDX_DETERMINISTIC=1 over a long run: N optimiser steps of the benchmark step (B = 6, 192x64x48) in a fresh process, then the
sha256 of all parameters and the last losses.  The parent runs the child twice per mode and compares the digests.
GPU box: python tools/det_soak.py [--steps 1000] [--modes bf16,fp16]"""
import argparse, hashlib, os, subprocess, sys
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1000); ap.add_argument("--modes", default="bf16,fp16"); ap.add_argument("--child", default=None)
a = ap.parse_args()
if a.child is None:
    for mode in a.modes.split(","):
        for det in ("1", "0"):
            env = dict(os.environ, TDX_DETERMINISTIC=det)
            outs = [subprocess.run([sys.executable, __file__, "--steps", str(a.steps), "--child", mode], env=env, capture_output=True,
                                   text=True).stdout.strip().splitlines()[-1] for _ in range(2)]
            print(f"{mode} TDX_DETERMINISTIC={det}, {a.steps} steps, two processes: {'IDENTICAL' if outs[0].split("  (")[0] == outs[1].split("  (")[0] else 'differ'}\n   {outs[0]}\n   {outs[1]}",
                  flush=True)
    sys.exit(0)

import time
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning

dev = torch.device("cuda:0")
torch.manual_seed(0)
diff = bench.build_model(dev)
bench.set_mode(diff, a.child)
x, c, idx = bench.synthetic_inputs(6, dev)
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
opt = bench.new_optimizer(diff, a.child, bench.LOSS_ELEMENTS(6, idx))
torch.manual_seed(0)
losses = []
torch.cuda.synchronize(); t0 = time.time()
for step in range(a.steps):
    loss, _ = diff(x, C, md, None)
    opt.scale_loss(loss).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    if step >= a.steps - 3:
        losses.append(loss.detach())
torch.cuda.synchronize()
ms = 1e3 * (time.time() - t0) / a.steps
h = hashlib.sha256()
for n, p in diff.model.named_parameters():
    h.update(p.detach().cpu().numpy().tobytes())
print(f"params sha256 {h.hexdigest()[:16]}  last losses {[round(l.item(), 9) for l in losses]}  ({ms:.2f} ms per step)")
