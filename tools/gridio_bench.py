#!/usr/bin/env python3
"""Ingress / egress kernels at BASELINE configs[1] (192 x 64 x 48, variables (u, p), B = 6) against the
reference's own op chain run through stock PyTorch-ROCm on the same GPU, and against the HBM roofline.
Usage (GPU box): python tools/gridio_bench.py [--batch 6]"""
import argparse, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "generative-turbulence_amd")); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from turbdiff_amd import gridio
from turbdiff_amd.data.ofles import Variable
from turbdiff_amd.models.cell_type_embeddings import CellTypeEmbedding
from turbdiff_amd.models.normalization import Normalization
from grid_cases import full_size_batch as _full_size_batch

HBM = 8e12


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=6); a = ap.parse_args()
    batch = _full_size_batch(B=a.batch); data, meta, stats = batch.data, batch.data.metadata, batch.stats
    vs = (Variable.U, Variable.P); mode = "u:norm-max;p:abs-max"
    norm = Normalization(vs, mode); B, F = a.batch, 4
    V, n = int(np.prod(meta.cell_counts)), meta.n_cells
    mean, std = stats.normalizers(vs, mode); mean, std = mean.cuda(), std.cuda()
    walls, inlets, outlets = (meta.boundaries[k]["idx"] for k in ("walls", "inlets", "outlets"))
    inlet_v = torch.tensor([1.0, 0.0, 0.0], device="cuda")[:, None]

    def ref_ingress():  # ofles.py:220-240 + normalization.py:19-23 as the reference issues them
        x = torch.zeros((B, F, V), device="cuda")
        x[:, :3, meta.cell_idx] = data.samples[Variable.U].transpose(1, 2)
        x[:, 3:, meta.cell_idx] = data.samples[Variable.P].transpose(1, 2)
        x[:, :3, walls] = 0.0; x[:, :3, inlets] = inlet_v; x[:, 3:, outlets] = 0.0
        return torch.addcmul((-mean / std).view(-1, 1), torch.reciprocal(std).view(-1, 1), x)

    xn = norm.normalized_grid_embedding(data, stats)
    assert torch.allclose(xn.flatten(2), ref_ingress(), atol=1e-6)

    def ref_egress():  # normalization.py:25-29 + utils.py:14-15 + metrics.py:52-58 (without the .cpu())
        y = torch.addcmul(mean.view(-1, 1, 1, 1), std.view(-1, 1, 1, 1), xn).flatten(start_dim=-3)[..., meta.cell_idx]
        y = y.permute(0, 2, 1)
        return [c.contiguous() for c in torch.split(y, [3, 1], dim=-1)]

    emb = CellTypeEmbedding.create("learned", 4).cuda()
    plan = gridio.plan_for(meta); types64 = plan.types.long().reshape(plan.counts)
    gC = torch.randn(4, *plan.counts, device="cuda")

    def ours_embed_fb():
        C = emb(data); return torch.autograd.grad(C, emb.embedding.weight, gC)

    def ref_embed_fb():
        C = torch.movedim(emb.embedding(types64), -1, 0); return torch.autograd.grad(C, emb.embedding.weight, gC)

    rows = [
        ("ingress: grid_embedding + normalize_grid", lambda: norm.normalized_grid_embedding(data, stats), ref_ingress,
         B * F * V * 4 + B * n * F * 4 + V * 8),
        ("egress: denormalize + select_cells + channels-last", lambda: norm.denormalized_cells(xn, meta, stats), ref_egress,
         2 * B * n * F * 4 + n * 8),
        ("cell-type embedding fwd + table gradient", ours_embed_fb, ref_embed_fb, 2 * 4 * V * 4 + 2 * V),
    ]
    print(f"B = {B}, grid {tuple(plan.counts)}, {n} in-domain cells of {V}; HBM roofline {HBM/1e12:.0f} TB/s")
    print(f"{'step':52s} {'HIP us':>8s} {'GB/s':>7s} {'% HBM':>6s} {'torch-ROCm us':>14s} {'speed-up':>8s}")
    for name, ours, ref, nbytes in rows:
        t, tr = timeit(ours), timeit(ref)
        print(f"{name:52s} {t*1e6:8.1f} {nbytes/t/1e9:7.0f} {100*nbytes/t/HBM:6.1f} {tr*1e6:14.1f} {tr/t:8.1f}x")


if __name__ == "__main__":
    main()
