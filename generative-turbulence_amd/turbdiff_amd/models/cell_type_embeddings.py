"""Cell-type conditioning (interface of turbdiff/models/cell_type_embeddings.py:14-87)."""

from __future__ import annotations

from typing import Literal

import torch
import torch.nn.functional as F
from torch import nn

from .. import gridio


class CellTypeEmbedding(nn.Module):
    """Mark each cell of a 3D grid with an embedding of its type."""

    @staticmethod
    def create(type: Literal["learned", "onehot"], dim: int):
        if type == "learned":
            return CellTypeLearnedEmbedding(dim)
        if type == "onehot":
            return CellTypeOneHotEmbedding()
        raise RuntimeError(f"Unknown cell type embedding {type}")

    def __init__(self):
        super().__init__()
        self.boundary_types = dict(gridio.CELL_TYPES)

    @property
    def n_types(self):
        return len(self.boundary_types)

    @property
    def out_dim(self):
        raise NotImplementedError()

    def _plan(self, data):
        return gridio.plan_for(data.metadata)

    def cell_types(self, data) -> torch.Tensor:
        plan = self._plan(data)
        return plan.types.long().reshape(plan.counts)


class CellTypeLearnedEmbedding(CellTypeEmbedding):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.embedding = nn.Embedding(self.n_types, embedding_dim=dim)

    def forward(self, data):
        return gridio.cell_type_embedding(self.embedding.weight, self._plan(data))

    @property
    def out_dim(self):
        return self.dim


class CellTypeOneHotEmbedding(CellTypeEmbedding):
    def forward(self, data):
        return torch.movedim(F.one_hot(self.cell_types(data), num_classes=self.n_types), -1, 0)

    @property
    def out_dim(self):
        return self.n_types
