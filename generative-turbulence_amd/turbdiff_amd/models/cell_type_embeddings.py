"""Cell-type conditioning on the device.

Interface of the reference's ``turbdiff/models/cell_type_embeddings.py`` (classes ``CellTypeEmbedding`` with
its ``create`` factory, ``CellTypeLearnedEmbedding``, ``CellTypeOneHotEmbedding``; attributes ``boundary_types``,
``n_types``, ``out_dim``, ``embedding``; methods ``cell_types`` and ``forward``), built on the per-geometry
plan of ``turbdiff_amd.gridio``: the type grid is computed once per geometry as ``uint8`` (reference lines
47-59 recompute an int64 grid per call), the learned variant looks its table up with ``tdx_cell_embed_fwd`` and
gets the table gradient from ``tdx_cell_embed_bwd`` (reference lines 69-70: ``nn.Embedding`` + ``movedim``).
"""

from __future__ import annotations

import torch
from torch import nn

from .. import gridio

_KINDS = ("learned", "onehot")


class CellTypeEmbedding(nn.Module):
    """Base of the two variants; ``CellTypeEmbedding.create(kind, dim)`` builds one."""

    def __init__(self):
        super().__init__()
        self.boundary_types = dict(gridio.CELL_TYPES)  # name -> integer label, as in the reference

    @staticmethod
    def create(type: str, dim: int) -> "CellTypeEmbedding":
        if type not in _KINDS:
            raise RuntimeError(f"Unknown cell type embedding {type}")
        return CellTypeLearnedEmbedding(dim) if type == "learned" else CellTypeOneHotEmbedding()

    # -- what the model asks ------------------------------------------------------------------------------
    n_types = property(lambda self: len(self.boundary_types))

    @property
    def out_dim(self) -> int:
        raise NotImplementedError()

    # -- geometry -------------------------------------------------------------------------------------------
    def _plan(self, data) -> gridio.GridPlan:
        return gridio.plan_for(data.metadata)

    def cell_types(self, data) -> torch.Tensor:
        """(X, Y, Z) int64 labels: outside, inside at the cells, then every boundary by name."""
        plan = self._plan(data)
        return plan.types.to(torch.int64).view(plan.counts)


class CellTypeLearnedEmbedding(CellTypeEmbedding):
    """A trainable (n_types, dim) table; the state_dict key stays ``embedding.weight``."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = int(dim)
        self.embedding = nn.Embedding(num_embeddings=self.n_types, embedding_dim=self.dim)

    out_dim = property(lambda self: self.dim)

    def forward(self, data) -> torch.Tensor:
        # (dim, X, Y, Z), differentiable with respect to the table
        return gridio.cell_type_embedding(self.embedding.weight, self._plan(data))


class CellTypeOneHotEmbedding(CellTypeEmbedding):
    """One channel per type (int64, like ``F.one_hot``); no parameters, plain torch indexing."""

    out_dim = property(lambda self: self.n_types)

    def forward(self, data) -> torch.Tensor:
        labels = self.cell_types(data)
        eye = torch.eye(self.n_types, dtype=torch.int64, device=labels.device)
        return eye[:, labels]
