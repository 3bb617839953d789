#!/usr/bin/env python3
"""Training-trajectory comparison of the three numeric modes on the same seeds: 150 optimiser steps of the
BASELINE configs[1] step (B = 6, 192x64x48) in fp32 (IEEE fp32 MFMA convs), f32s (fp32 tensors, split-precision
convs) and bf16; prints the loss every 10 steps, the largest relative loss deviation from the fp32 run, time
per step and peak memory.  GPU box: python tools/soak_train.py [--steps 150] [--modes f32,f32s,bf16]"""
import argparse, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
from types import SimpleNamespace
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.optim import ClipRAdam


def run(mode, steps):
    os.environ.pop("TDX_CONV_IMPL", None)
    if mode == "f32s":
        os.environ["TDX_CONV_IMPL"] = "split"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    diff = bench.build_model(dev, torch.bfloat16 if mode == "bf16" else torch.float32)
    opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1)
    x, c, idx = bench.synthetic_inputs(6, dev)
    C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
    torch.manual_seed(0)  # same t and noise draws in every mode
    torch.cuda.reset_peak_memory_stats()
    losses = []
    torch.cuda.synchronize(); t0 = time.time()
    for step in range(steps):
        loss, _ = diff(x, C, md, None)
        loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(loss.detach())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps * 1e3
    return [l.item() for l in losses], dt, torch.cuda.max_memory_allocated() / 1e9


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--modes", default="f32,f32s,bf16"); a = ap.parse_args()
    ref = None
    for mode in a.modes.split(","):
        losses, dt, mem = run(mode, a.steps)
        line = f"{mode:5s} {dt:7.1f} ms/step  peak {mem:5.1f} GB  loss[::10] " + " ".join(f"{l:.4f}" for l in losses[::10])
        if ref is None:
            ref = losses
        else:
            dev_ = max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(losses, ref))
            line += f"   max |dloss|/loss vs {a.modes.split(',')[0]}: {dev_:.2e}"
        assert all(l == l for l in losses), "non-finite loss"
        print(line, flush=True)


if __name__ == "__main__":
    main()
