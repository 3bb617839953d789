"""Cell-index helpers with the reference's names and semantics (turbdiff/models/utils.py:8-28).

These are the torch-level forms kept for callers outside the hot path (sample stores,
metrics).  Inside the hot path the flat ``cell_idx`` list is turned once into a dense uint8
mask (``ops.cell_mask``) that the fused HIP kernels consume.
"""

from __future__ import annotations

import torch

_GRID_DIMS = 3  # the trailing (X, Y, Z) axes of every field tensor


def ravel_cells(x: torch.Tensor) -> torch.Tensor:
    """(..., X, Y, Z) -> (..., X*Y*Z): a view for contiguous inputs, so writes go through."""
    return x.flatten(start_dim=-_GRID_DIMS)


def select_cells(x: torch.Tensor, cell_idx: torch.Tensor) -> torch.Tensor:
    """Values at the flat in-domain indices: (..., X, Y, Z) -> (..., n_cells)."""
    return torch.index_select(ravel_cells(x), -1, cell_idx)


def where_cells(cell_idx: torch.Tensor, cell_values: torch.Tensor, other: torch.Tensor | None = None) -> torch.Tensor:
    """`cell_values` at the in-domain cells `cell_idx`, `other` (or zero) everywhere else; inputs untouched."""
    result = (other.clone(memory_format=torch.contiguous_format) if other is not None
              else cell_values.new_zeros(cell_values.shape))
    ravel_cells(result).index_copy_(-1, cell_idx, select_cells(cell_values, cell_idx))
    return result


def broadcast_right(x: torch.Tensor, other: torch.Tensor) -> torch.Tensor:
    """Append singleton dims to `x` until it has `other`'s rank (per-sample scalars against fields)."""
    missing = other.ndim - x.ndim
    if missing < 0:
        raise AssertionError("`other` must have at least as many dimensions as `x`")
    return x[(...,) + (None,) * missing]
