#!/usr/bin/env python3
"""Training-trajectory comparison of the numeric modes on the same seeds: 150 optimiser steps of the
BASELINE configs[1] step (B = 6, 192x64x48) in fp32 (IEEE fp32 MFMA convs), f32s (fp32 tensors, split-precision
convs), bf16 and fp16 (loss-scaled); prints the loss every 10 steps, the largest relative loss deviation from the first
mode's run, time per step, peak memory and (fp16) the loss scale / skipped steps.
GPU box: python tools/soak_train.py [--steps 150] [--modes f32,f32s,bf16,fp16]"""
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
    diff = bench.build_model(dev, bench.MODE_DTYPE[mode])
    x, c, idx = bench.synthetic_inputs(6, dev)
    opt = bench.new_optimizer(diff, mode, bench.LOSS_ELEMENTS(6, idx))
    C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
    torch.manual_seed(0)  # same t and noise draws in every mode
    torch.cuda.reset_peak_memory_stats()
    losses = []
    torch.cuda.synchronize(); t0 = time.time()
    for step in range(steps):
        loss, _ = diff(x, C, md, None)
        opt.scale_loss(loss).backward(); opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(loss.detach())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps * 1e3
    opt.settle()
    run.note = f"  loss scale 2^{int(torch.tensor(opt.loss_scale).log2())}, {opt.skipped_steps} skipped" if opt.loss_scale else ""
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
        print(line + run.note, flush=True)


if __name__ == "__main__":
    main()
