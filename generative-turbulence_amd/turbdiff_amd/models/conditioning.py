"""Conditioning dictionary helpers (interface of turbdiff/models/conditioning.py:13-93).

``C`` maps a conditioning type to an unbatched (c, X, Y, Z) tensor.  The keys only need a
boolean ``local`` / ``global_`` attribute, so the reference's own ``Conditioning.Type`` enum
members work unchanged when this package is used as a drop-in.
"""

import enum

import torch
from torch import nn


class Conditioning(nn.Module):
    """conditioning.py:14-83: builds the conditioning dictionary of a batch (cell-type embedding and / or
    normalised cell positions)."""

    def __init__(self, variables=(), cell_type_embedding=None, cell_pos: bool = False):
        super().__init__()
        self.variables = variables
        self.cell_type_embedding = cell_type_embedding
        self.cell_pos = cell_pos

    def forward(self, data):
        C = {}
        if self.cell_type_embedding is not None:
            C[Conditioning.Type.CELL_TYPE] = self.cell_type_embedding(data)
        if self.cell_pos:
            axes = [torch.linspace(0, 1, int(c), device=data.device) for c in data.metadata.cell_counts]
            C[Conditioning.Type.CELL_POS] = torch.stack(torch.meshgrid(*axes, indexing="ij"))
        return C

    @property
    def local_conditioning_dim(self):
        dim = self.cell_type_embedding.out_dim if self.cell_type_embedding is not None else 0
        return dim + (3 if self.cell_pos else 0)

    @property
    def global_conditioning_dim(self):
        return 0

    class Type(enum.Enum):
        CELL_TYPE = enum.auto()
        CELL_POS = enum.auto()

        @property
        def local(self):
            return True

        @property
        def global_(self):
            return False


def _gather(C, attr):
    parts = [v for k, v in C.items() if getattr(k, attr)]
    if len(parts) == 1:
        return parts[0]  # (torch.cat of ONE tensor copies it: 9.4 MB of cell types per forward at 192x64x48, a memcpy node in a captured step)
    return torch.cat(parts, dim=0) if parts else None


def local_conditioning(C):
    """Channel concatenation (dict order) of the per-cell conditionings, or None."""
    return _gather(C, "local")


def global_conditioning(C):
    return _gather(C, "global_")
