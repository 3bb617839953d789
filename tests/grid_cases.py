"""Loader for the cases of tests/golden/grid_io.npz (made by tests/golden/make_golden_grid.py)."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_case(tag):
    z = np.load(GOLDEN / "grid_io.npz")
    variables = tuple((s.split(":")[0], int(s.split(":")[1])) for s in z[f"{tag}/variables"])
    boundaries = {str(k): z[f"{tag}/boundary/{k}"] for k in z[f"{tag}/boundary_order"]}
    fixed = {}
    for name, _ in variables:
        key = f"{tag}/fixed_order/{name}"
        if key in z.files:
            fixed[name] = {str(b): z[f"{tag}/fixed/{name}/{b}"] for b in z[key]}
    stats = {}
    for k in z.files:
        if k.startswith(f"{tag}/stats/"):
            _, _, key, sn = k.split("/")
            stats.setdefault(key, {})[sn] = z[k]
    modes = {}
    for k in z.files:
        parts = k.split("/")
        if len(parts) == 3 and parts[0] == tag and parts[1] == "mode":
            mi = parts[2]
            modes[str(z[k])] = SimpleNamespace(mean=z[f"{tag}/mode/{mi}/mean"], std=z[f"{tag}/mode/{mi}/std"],
                                               normalized=z[f"{tag}/mode/{mi}/normalized"],
                                               denormalized=z[f"{tag}/mode/{mi}/denormalized"])
    return SimpleNamespace(
        tag=tag, z=z, variables=variables, cell_counts=tuple(int(c) for c in z[f"{tag}/cell_counts"]),
        cell_idx=z[f"{tag}/cell_idx"], boundaries=boundaries, fixed=fixed, stats=stats, modes=modes,
        samples={name: z[f"{tag}/samples/{name}"] for name, _ in variables},
        grid_embedding=z[f"{tag}/grid_embedding"], cell_types=z[f"{tag}/cell_types"],
        table=z[f"{tag}/embedding/table"], learned=z[f"{tag}/embedding/learned"], onehot=z[f"{tag}/embedding/onehot"],
        grad_out=z[f"{tag}/embedding/grad_out"], grad_table=z[f"{tag}/embedding/grad_table"],
        egress_x=z[f"{tag}/egress/x"], egress={name: z[f"{tag}/egress/{name}"] for name, _ in variables})


def full_size_batch(B=2, seed=0):
    """BASELINE configs[1] geometry: 192 x 64 x 48, interior minus a 12 x 16 x 32 obstacle."""
    from turbdiff_amd.data.ofles import (BoundaryCondition, OpenFOAMBatch, OpenFOAMData, OpenFOAMMetadata, OpenFOAMStats,
                                         Variable)

    g = torch.Generator().manual_seed(seed)
    n = (192, 64, 48)
    inside = torch.zeros(n, dtype=torch.bool)
    inside[1:-1, 1:-1, 1:-1] = True
    inside[13:25, 1:17, 1:33] = False
    idx = torch.arange(inside.numel()).reshape(n)
    cell_idx = idx[inside]
    walls = torch.cat([idx[:, 0, :].flatten(), idx[:, -1, :].flatten(), idx[:, :, 0].flatten(), idx[:, :, -1].flatten(),
                       idx[13:25, 1:17, 1:33].flatten()])
    boundaries = {"walls": {"idx": walls.to("cuda:0")}, "inlets": {"idx": idx[0].flatten().to("cuda:0")},
                  "outlets": {"idx": idx[-1].flatten().to("cuda:0")}}
    FV = BoundaryCondition.Type.FIXED_VALUE
    bcs = {Variable.U: {"walls": BoundaryCondition(FV, torch.zeros(3)), "inlets": BoundaryCondition(FV, torch.tensor([1.0, 0.0, 0.0])),
                        "outlets": BoundaryCondition(BoundaryCondition.Type.INLET_OUTLET)},
           Variable.P: {"outlets": BoundaryCondition(FV, torch.tensor(0.0))}}
    meta = OpenFOAMMetadata(np.array(n), cell_idx.to("cuda:0"), boundaries, bcs)
    samples = {Variable.U: torch.randn(B, len(cell_idx), 3, generator=g).to("cuda:0"),
               Variable.P: torch.randn(B, len(cell_idx), 1, generator=g).to("cuda:0")}
    stats = OpenFOAMStats({"u": {"mean": torch.tensor([0.3, 0.0, 0.1]), "std": torch.tensor([0.5, 0.2, 0.25]),
                                 "min": -torch.ones(3), "max": torch.ones(3) * 2},
                           "p": {"mean": torch.tensor(0.1), "std": torch.tensor(0.7), "min": torch.tensor(-1.0), "max": torch.tensor(3.0)},
                           "norm(u)": {"mean": torch.tensor(0.9), "std": torch.tensor(0.4), "min": torch.tensor(0.0), "max": torch.tensor(2.5)}})
    return OpenFOAMBatch(OpenFOAMData(meta, torch.zeros(B), samples), stats)


