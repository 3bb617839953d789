"""Loader for the cases of tests/golden/grid_io.npz (made by tests/golden/make_golden_grid.py)."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np

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
