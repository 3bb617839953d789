#!/usr/bin/env python3
"""Captured training step against the eager one over many optimiser steps, bit for bit: with TDX_DETERMINISTIC=1 every merge of the
backward pass is ordered, so N steps on the same (x, t, noise) from the same weights must end in the SAME parameters whether forward +
backward are issued eagerly or replayed from one captured graph (weights re-packed inside the graph, clip + RAdam eager in both).
GPU box: python tools/graph_vs_eager_soak.py [--steps 100] [--mode bf16] [--grid 96 32 24] [--batch 2]"""
import argparse, hashlib, os, sys
from pathlib import Path
from types import SimpleNamespace
os.environ.setdefault("TDX_DETERMINISTIC", "1")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.training import GraphedTrainingStep

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100); ap.add_argument("--mode", default="bf16")
ap.add_argument("--grid", type=int, nargs=3, default=[96, 32, 24]); ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
diff = bench.build_model(dev)
bench.set_mode(diff, a.mode)
sd0 = {k: v.clone() for k, v in diff.state_dict().items()}
x, c, idx = bench.synthetic_inputs(a.batch, dev, tuple(a.grid))
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
t = torch.tensor(([3, 250, 499, 17, 120, 380] * a.batch)[: a.batch], device=dev)
noise = torch.randn(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))


def digest():
    h = hashlib.sha256()
    for n, p in diff.model.named_parameters():
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


out = {}
for kind in ("eager", "graph"):
    diff.load_state_dict(sd0)
    diff.zero_grad(set_to_none=True)
    opt = bench.new_optimizer(diff, a.mode, bench.LOSS_ELEMENTS(a.batch, idx))
    losses = []
    if kind == "graph":
        gs = GraphedTrainingStep(bench._Task(diff, opt), inject=True)
        gs.set_draws(t, noise)
        batch = SimpleNamespace(x=x, C=C, cell_idx=idx)
    for step in range(a.steps):
        if kind == "graph":
            loss = gs(batch)
        else:
            opt.zero_grad(set_to_none=True)
            loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
            opt.scale_loss(loss).backward()
        opt.step()
        losses.append(loss.detach().clone())
        del loss  # (an eager step's autograd graph must not be alive when the capture begins)
    torch.cuda.synchronize()
    out[kind] = (digest(), [round(l.item(), 9) for l in losses[-3:]])
    print(kind, out[kind], flush=True)
print("IDENTICAL" if out["eager"] == out["graph"] else "DIFFERENT")
