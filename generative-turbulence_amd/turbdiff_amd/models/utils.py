"""Cell-index helpers with the reference's names and semantics (turbdiff/models/utils.py:8-28).

These are the torch-level forms kept for callers outside the hot path (sample stores,
metrics).  Inside the hot path the flat ``cell_idx`` list is turned once into a dense uint8
mask (``ops.cell_mask``) that the fused HIP kernels consume.
"""

import torch


def broadcast_right(x: torch.Tensor, other: torch.Tensor):
    """Append singleton dims to `x` until it broadcasts against `other` from the left."""
    assert other.ndim >= x.ndim
    return x.reshape(x.shape + (1,) * (other.ndim - x.ndim))


def ravel_cells(x: torch.Tensor):
    return x.flatten(start_dim=-3)


def select_cells(x: torch.Tensor, cell_idx: torch.Tensor):
    return ravel_cells(x)[..., cell_idx]


def where_cells(cell_idx, cell_values, other: torch.Tensor | None = None):
    """`cell_values` at the in-domain cells `cell_idx`, `other` (or zero) everywhere else."""
    out = torch.zeros_like(cell_values) if other is None else other.clone()
    ravel_cells(out)[..., cell_idx] = ravel_cells(cell_values)[..., cell_idx]
    return out
