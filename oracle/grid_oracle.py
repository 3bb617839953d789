"""CPU oracle for the data ingress / egress either side of the hot path.
TEST INFRASTRUCTURE ONLY.

A restatement (numpy + stock PyTorch-CPU ops, fp32) of what the reference does between the HDF5 sample
arrays and the tensors ``GaussianDiffusion`` sees, and back:

* ``OpenFOAMData.grid_embedding``          turbdiff/data/ofles.py:220-240
* ``OpenFOAMStats.normalizers``            turbdiff/data/ofles.py:249-293
* ``Normalization.(de)normalize_grid``     turbdiff/models/normalization.py:19-41
* ``CellTypeEmbedding.cell_types`` and the learned / one-hot embeddings
                                           turbdiff/models/cell_type_embeddings.py:47-83
* ``select_cells`` + channels-last split as ``SampleStore.add_samples`` prepares it
                                           turbdiff/models/utils.py:14-15, turbdiff/models/metrics.py:50-58

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/grid_io.npz``, which ``tests/golden/make_golden_grid.py`` produced by running the
unmodified reference classes in the build container.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs may import this module.

Conventions: ``variables`` is a tuple of ``(name, dims)`` pairs, e.g. ``(("u", 3), ("p", 1))``;
``boundaries`` an ordered mapping name -> int64 index array; ``fixed_values`` maps variable name ->
ordered mapping boundary name -> value array (only the FIXED_VALUE conditions; the order is the
reference's iteration order of ``boundary_conditions[v]``).
"""

from __future__ import annotations

import numpy as np
import torch

CELL_TYPES = {"inside": 0, "outside": 1, "walls": 2, "inlets": 3, "outlets": 4, "empties": 5}  # cell_type_embeddings.py:30-38


def grid_embedding(samples, variables, cell_idx, cell_counts, boundaries, fixed_values):
    """ofles.py:220-240.  samples[name]: (B, n_cells, dims) fp32 -> (B, sum dims, X, Y, Z) fp32."""
    first = np.asarray(samples[variables[0][0]])
    B = first.shape[0]
    F = sum(d for _, d in variables)
    V = int(np.prod(cell_counts))
    x = np.zeros((B, F, V), dtype=np.float32)
    cell_idx = np.asarray(cell_idx)
    f0 = 0
    for name, d in variables:  # ofles.py:231-232: scatter the channels-last samples
        x[:, f0:f0 + d, cell_idx] = np.transpose(np.asarray(samples[name], dtype=np.float32), (0, 2, 1))
        f0 += d
    f0 = 0
    for name, d in variables:  # ofles.py:235-238: FIXED_VALUE boundary conditions, in iteration order
        for bname, value in fixed_values.get(name, {}).items():
            idx = np.asarray(boundaries[bname])
            x[:, f0:f0 + d, idx] = np.broadcast_to(np.asarray(value, dtype=np.float32).reshape(-1, 1), (d, len(idx)))
        f0 += d
    return x.reshape(B, F, *[int(c) for c in cell_counts])


def normalizers(stats, variables, mode):
    """ofles.py:249-293.  stats[key][stat] are arrays; returns (mean, std) fp32 of length sum dims."""
    if ":" in mode:
        per_var = {cfg.split(":")[0].lower(): cfg.split(":")[1] for cfg in mode.split(";")}
        mode_of = lambda name: per_var[name]  # noqa: E731
    else:
        mode_of = lambda name: mode  # noqa: E731
    mean, std = [], []
    for name, d in variables:
        m, s = np.zeros(d, dtype=np.float32), np.ones(d, dtype=np.float32)
        vm = mode_of(name)
        if "norm" in vm:
            st = stats[f"norm({name})"]
            if vm == "norm":
                s[:] = st["mean"]
            elif vm == "norm-std":
                m[:] = st["mean"]
                s[:] = st["std"]
            elif vm == "norm-max":
                s[:] = st["max"]
            else:
                raise RuntimeError(f"Unknown normalization mode {vm}")
        else:
            st = stats[name]
            if vm == "abs-max":
                s[:] = np.maximum(np.abs(st["min"]), np.abs(st["max"]))
            elif vm == "mean-std":
                m[:] = st["mean"]
                s[:] = st["std"]
            elif vm == "std":
                s[:] = st["std"]
            else:
                raise RuntimeError(f"Unknown normalization mode {vm}")
        mean.append(m)
        std.append(s)
    mean, std = np.concatenate(mean), np.concatenate(std)
    std = np.where(std >= 1e-8, std, np.float32(1.0)).astype(np.float32)  # ofles.py:291
    return mean, std


def normalize_grid(x, mean, std):
    """normalization.py:19-23: addcmul(-mean/std, 1/std, x), fp32, broadcast over (X, Y, Z)."""
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1, 1)
    return torch.addcmul(-m / s, torch.reciprocal(s), torch.as_tensor(x)).numpy()


def denormalize_grid(x, mean, std):
    """normalization.py:25-29: addcmul(mean, std, x)."""
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1, 1)
    return torch.addcmul(m, s, torch.as_tensor(x)).numpy()


def cell_types(cell_idx, cell_counts, boundaries):
    """cell_type_embeddings.py:47-59: outside everywhere, inside at the cells, then every boundary by name."""
    t = np.full(int(np.prod(cell_counts)), CELL_TYPES["outside"], dtype=np.int64)
    t[np.asarray(cell_idx)] = CELL_TYPES["inside"]
    for name, idx in boundaries.items():
        t[np.asarray(idx)] = CELL_TYPES[name]
    return t.reshape([int(c) for c in cell_counts])


def cell_type_embedding(types, table):
    """cell_type_embeddings.py:69-70: movedim(embedding(types), -1, 0) -> (D, X, Y, Z)."""
    return np.moveaxis(np.asarray(table)[np.asarray(types)], -1, 0)


def cell_type_embedding_grad(types, dC, n_types=6):
    """Gradient of the table: dE[k, d] = sum over voxels of type k of dC[d, voxel] (fp64 accumulate)."""
    types = np.asarray(types).reshape(-1)
    dC = np.asarray(dC, dtype=np.float64).reshape(dC.shape[0], -1)
    out = np.zeros((n_types, dC.shape[0]))
    for k in range(n_types):
        out[k] = dC[:, types == k].sum(axis=1)
    return out


def cell_type_onehot(types, n_types=6):
    """cell_type_embeddings.py:78-81."""
    return np.moveaxis(np.eye(n_types, dtype=np.int64)[np.asarray(types)], -1, 0)


def select_cells_channels_last(x, cell_idx, variables):
    """utils.py:14-15 + metrics.py:52-58: per variable the (B, n_cells, dims) arrays SampleStore writes."""
    x = np.asarray(x)
    flat = x.reshape(*x.shape[:-3], -1)[..., np.asarray(cell_idx)]  # (B, F, n_cells)
    cl = np.transpose(flat, (0, 2, 1))
    out, f0 = {}, 0
    for name, d in variables:
        out[name] = np.ascontiguousarray(cl[..., f0:f0 + d])
        f0 += d
    return out
