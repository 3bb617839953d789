"""Per-feature normalisation (interface of turbdiff/models/normalization.py:12-41) plus the fused
ingress / egress of the hot path."""

from __future__ import annotations

import torch
from torch import nn

from .. import gridio


class Normalization(nn.Module):
    def __init__(self, variables, mode: str):
        super().__init__()
        self.variables = variables
        self.mode = mode

    # ---- the reference's two methods, on dense tensors (elementwise glue; any device)
    def normalize_grid(self, x: torch.Tensor, stats):
        mean, std = self._mean_and_std_3d(stats, x.device)
        return torch.addcmul(-mean / std, torch.reciprocal(std), x)

    def denormalize_grid(self, x: torch.Tensor, stats):
        mean, std = self._mean_and_std_3d(stats, x.device)
        return torch.addcmul(mean, std, x)

    def _mean_and_std_3d(self, stats, device=None):
        mean, std = stats.normalizers(self.variables, self.mode)
        return mean.to(device).view(-1, 1, 1, 1), std.to(device).view(-1, 1, 1, 1)

    # ---- fused with the sparse <-> dense conversion (one HIP kernel each)
    def normalized_grid_embedding(self, data, stats):
        """``normalize_grid(data.grid_embedding(variables), stats)`` (diffusion.py:238-239) in one pass."""
        mean, std = stats.normalizers(self.variables, self.mode)
        return gridio.grid_embed(data, self.variables, mean, std)

    def denormalized_cells(self, x: torch.Tensor, metadata, stats):
        """Per variable the (B, n_cells, dims) channels-last values of ``denormalize_grid(x)`` at the in-domain
        cells -- what ``SampleStore.add_samples`` stores (diffusion.py:157, metrics.py:52-58)."""
        mean, std = stats.normalizers(self.variables, self.mode)
        return gridio.grid_select(x, metadata, self.variables, mean, std)
