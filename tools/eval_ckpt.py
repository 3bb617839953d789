#!/usr/bin/env python3
"""Evaluate a checkpoint: counterpart of the reference's scripts/eval_ckpt.py:43-76 on the MI355X build.

    python tools/eval_ckpt.py CKPT SAMPLES.npz [--cases CASES.pt | --synthetic N] [key=value ...]

Flow, step for step as in the reference: load the checkpoint -> take the run configuration from ``ckpt["config"]``
-> merge dotted ``key=value`` overrides -> seed -> build the task from the configuration
(``DiffusionTrainer.from_config``; ``matmul_precision`` picks the compute mode) -> ``load_state_dict(strict)``
-> for every validation batch: ``task.sample`` -> ``SampleStore`` -> metrics over the stored samples.

Differences, all on the I/O side (no h5py / POT in this image): validation cases come from ``--cases`` (a
``torch.save``d list of ``(OpenFOAMMetadata, times, {Variable: (T, n_cells, dims)})`` tuples plus the statistics, see
``turbdiff_amd.data.ofles.InMemoryRepository``) or are synthesised; the store lives in memory and is written as one
``.npz``; of the reference's metrics the log-TKE-spectrum L2 distance runs here (on the device), the Wasserstein
ones need POT.
"""

import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "generative-turbulence_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def merge_overrides(config: dict, overrides):
    """``a.b.c=value`` items onto a nested dict (OmegaConf.merge(run_config, OmegaConf.from_cli(overrides)))."""
    import yaml

    for item in overrides:
        key, _, value = item.partition("=")
        node = config
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(value)
    return config


def synthetic_cases(n_cases, grid=(48, 32, 32), n_times=12, seed=0):
    """Channel-flow-like cases with a box obstacle: metadata, times and (u, p) samples at the in-domain cells."""
    from turbdiff_amd.data.ofles import BoundaryCondition, OpenFOAMMetadata, OpenFOAMStats, Variable

    g = torch.Generator().manual_seed(seed)
    X, Y, Z = grid
    cases = []
    for c in range(n_cases):
        inside = torch.zeros(grid, dtype=torch.bool)
        inside[1:-1, 1:-1, 1:-1] = True
        x0 = 6 + 3 * c
        inside[x0:x0 + 6, Y // 2 - 4:Y // 2 + 4, 0:Z // 2] = False
        cell_idx = inside.flatten().nonzero().flatten()
        flat = torch.arange(X * Y * Z).view(grid)
        boundaries = {"inlets": {"idx": flat[0].flatten()}, "outlets": {"idx": flat[-1].flatten()},
                      "walls": {"idx": torch.cat((flat[1:-1, 0].flatten(), flat[1:-1, -1].flatten()))}}
        bcs = {Variable.U: {"inlets": BoundaryCondition(BoundaryCondition.Type.FIXED_VALUE, torch.tensor([1.0, 0.0, 0.0])),
                            "walls": BoundaryCondition(BoundaryCondition.Type.FIXED_VALUE, torch.zeros(3)),
                            "outlets": BoundaryCondition(BoundaryCondition.Type.ZERO_GRADIENT)},
               Variable.P: {"inlets": BoundaryCondition(BoundaryCondition.Type.ZERO_GRADIENT),
                            "walls": BoundaryCondition(BoundaryCondition.Type.ZERO_GRADIENT),
                            "outlets": BoundaryCondition(BoundaryCondition.Type.FIXED_VALUE, torch.zeros(1))}}
        meta = OpenFOAMMetadata(np.array(grid), cell_idx, boundaries, bcs, file=Path(f"case-{c:02d}/data.h5"))
        n = len(cell_idx)
        u = torch.randn(n_times, n, 3, generator=g) * 0.3 + torch.tensor([1.0, 0.0, 0.0])
        p = torch.randn(n_times, n, 1, generator=g) * 0.1
        cases.append((meta, np.linspace(0.0, 1.1, n_times), {Variable.U: u, Variable.P: p}))
    allu = torch.cat([c[2][Variable.U].reshape(-1, 3) for c in cases])
    allp = torch.cat([c[2][Variable.P].reshape(-1, 1) for c in cases])
    rec = lambda t: {"mean": t.mean(0), "std": t.std(0), "min": t.amin(0), "max": t.amax(0)}
    nrm = lambda t: {k: v.reshape(()) for k, v in rec(t.norm(dim=-1, keepdim=True)).items()}
    stats = OpenFOAMStats({"u": rec(allu), "p": rec(allp), "norm(u)": nrm(allu), "norm(p)": nrm(allp)})
    return cases, stats


def evaluate(ckpt: dict, cases, stats, device, *, overrides=(), seed=2883413570083077179, samples_path=None,
             eval_batch_size=None, val_samples=2, discard_first_seconds=0.0, start_from=None, compute_mode=None):
    """The body of eval_ckpt.py:43-76.  Returns (store, metrics dict, task)."""
    from turbdiff_amd.data.ofles import InMemoryRepository, OpenFOAMDataset, OpenFOAMEvaluationSampler
    from turbdiff_amd.data.staging import DeviceStager
    from turbdiff_amd.models.metrics import LogTKESpectrumL2Distance, SampleStore
    from turbdiff_amd.training import DiffusionTrainer

    if "config" not in ckpt:
        raise SystemExit("Checkpoint has no config")
    config = merge_overrides(dict(ckpt["config"]), overrides)
    torch.manual_seed(seed % (2**63))
    task = DiffusionTrainer.from_config(config, max_train_steps=1, compute_mode=compute_mode)
    task.load_state_dict(ckpt["state_dict"], strict=True)
    task = task.to(device).eval()

    dataset = OpenFOAMDataset(InMemoryRepository(cases), stats, discard_first_seconds)
    bs = eval_batch_size or int(config["model"].get("eval_batch_size", 8))
    sampler = OpenFOAMEvaluationSampler(dataset, batch_size=bs, samples_per_file=val_samples)
    store = SampleStore(samples_path, task.variables)
    batches = (dataset[idxs] for idxs in sampler)
    with torch.no_grad():
        for batch in DeviceStager(batches, device):  # pinned staging + copy stream (move_data_to_device)
            if start_from is None:
                task.validation_step(batch, store)
            else:  # short runs: start the reverse process from a noised data sample
                store.add_cells(task.sample_cells(batch, start_from=start_from), batch.data.metadata)
    if samples_path is not None:
        store.save()

    # metric over the stored samples: log-TKE-spectrum L2 distance between generated and data velocity fields of
    # each case, on the largest obstacle-free box behind the inlet plane (metrics.py:319-380)
    from turbdiff_amd.data.ofles import Variable

    dist = LogTKESpectrumL2Distance(task.val_sample_metrics.metrics[0].distance.tke_spectrum).to(device)
    metrics = {}
    for meta, _, fields in cases:
        if meta.case_name not in store.case_names:
            continue
        gen = store.load_samples(meta).samples[Variable.U].to(device)       # (n, n_cells, 3)
        dat = fields[Variable.U][-gen.shape[0]:].to(device)
        X, Y, Z = (int(c) for c in meta.cell_counts)
        dense = lambda cells: torch.zeros(cells.shape[0], 3, X * Y * Z, device=device).index_copy_(
            2, meta.cell_idx.to(device), cells.transpose(1, 2)).view(-1, 3, X, Y, Z)
        box = (slice(None), slice(None), slice(X // 2, X - 1), slice(1, Y - 1), slice(1, Z - 1))
        ug, ud = dense(gen)[box].contiguous(), dense(dat)[box].contiguous()
        D, *_ = dist(ug, ud, ud.mean(0))
        metrics[f"val/{meta.case_name}/log_tke_l2"] = float(D.mean().item())
    if metrics:
        metrics["val/log_tke_l2"] = float(np.mean(list(metrics.values())))
    return store, metrics, task


def main():
    ap = argparse.ArgumentParser(description="Evaluate a checkpoint with overrides")
    ap.add_argument("-d", "--device", default="cuda")
    ap.add_argument("-s", "--seed", default=2883413570083077179, type=int)
    ap.add_argument("--cases", help="torch.save'd {'cases': [...], 'stats': OpenFOAMStats.stats} of the validation cases")
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic cases instead")
    ap.add_argument("--start-from", type=int, default=None, help="start the reverse process at this step (smoke runs)")
    ap.add_argument("--compute-mode", default=None, choices=[None, "f32", "f32s", "bf16", "fp16"])
    ap.add_argument("ckpt", help="Path to .ckpt file")
    ap.add_argument("samples_path", help=".npz file for storing samples")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_intermixed_args()  # options may sit between the positionals and the overrides
    samples_path = Path(args.samples_path)
    assert samples_path.suffix == ".npz" and not samples_path.exists()
    ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=False)
    if args.cases:
        from turbdiff_amd.data.ofles import OpenFOAMStats

        blob = torch.load(args.cases, weights_only=False)
        cases, stats = blob["cases"], OpenFOAMStats(blob["stats"])
    else:
        cases, stats = synthetic_cases(max(args.synthetic, 1))
    _, metrics, _ = evaluate(ckpt, cases, stats, torch.device(args.device), overrides=args.overrides, seed=args.seed,
                             samples_path=samples_path, start_from=args.start_from, compute_mode=args.compute_mode)
    for key in sorted(metrics):
        print(f"{key}: {metrics[key]}")


if __name__ == "__main__":
    main()
