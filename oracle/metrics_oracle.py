"""CPU oracle for the on-device sample metrics.  TEST INFRASTRUCTURE ONLY.

A restatement (stock PyTorch-CPU ops, fp32) of the reference's turbulent-kinetic-energy spectrum and the
log-spectrum distance built on it:

* ``interp3``                          turbdiff/models/metrics.py:220-268
* ``TurbulentKineticEnergySpectrum``   turbdiff/models/metrics.py:271-316
* ``LogTKESpectrumL2Distance``         turbdiff/models/metrics.py:319-380

Parity status: PINNED.  ``tests/test_metrics.py`` checks every function here against
``tests/golden/metrics.npz``, which ``tests/golden/make_golden_metrics.py`` produced by running the
unmodified reference classes in the build container (with the reference's own 5810-point Lebedev grid,
and with random quadrature points on odd grids).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs may import this module.
"""

from __future__ import annotations

import math

import torch


def interp3(grid: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """metrics.py:220-268: trilinear interpolation of (..., X, Y, Z) grids at (..., 3) points; the corner
    indices are clamped into the grid, the weights are taken relative to the CLAMPED lower corner."""
    hi = torch.tensor(grid.shape[-3:], dtype=torch.long) - 1
    p0 = torch.minimum(torch.clamp_min(torch.floor(points).long(), 0), hi)
    p1 = torch.minimum(torch.clamp_min(torch.floor(points).long() + 1, 0), hi)
    x0, y0, z0 = p0.unbind(-1)
    x1, y1, z1 = p1.unbind(-1)
    wx, wy, wz = (points - p0).unbind(-1)
    g = grid
    return ((1 - wx) * (1 - wy) * (1 - wz) * g[..., x0, y0, z0] + (1 - wx) * (1 - wy) * wz * g[..., x0, y0, z1]
            + (1 - wx) * wy * (1 - wz) * g[..., x0, y1, z0] + (1 - wx) * wy * wz * g[..., x0, y1, z1]
            + wx * (1 - wy) * (1 - wz) * g[..., x1, y0, z0] + wx * (1 - wy) * wz * g[..., x1, y0, z1]
            + wx * wy * (1 - wz) * g[..., x1, y1, z0] + wx * wy * wz * g[..., x1, y1, z1])


def tke_spectrum(u: torch.Tensor, k: torch.Tensor, p: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """metrics.py:289-316.  u (..., 3, X, Y, Z) velocity perturbation, k (K,) radii, (p, w) a quadrature
    rule on the unit sphere with weights summing to 1 -> E(k) of shape (..., K)."""
    tke = 0.5 * (u ** 2).sum(dim=-4)
    f = torch.fft.fftshift(torch.fft.fftn(tke, dim=(-3, -2, -1)), dim=(-3, -2, -1))
    center = k.new_tensor([s // 2 for s in u.shape[-3:]])
    q = k[:, None, None] * p + center
    val = interp3((f.abs() ** 2).log(), q).exp().float()
    return torch.matmul(val, w) * (4 * math.pi * k ** 2)


def log_tke_distance(u_a, u_b, u_mean, p, w, nodes, weights):
    """metrics.py:349-380: pairwise L2 distances of log E(k) with Gauss-Legendre nodes on [1, k_max]."""
    k_min, k_max = 1.0, float((min(u_a.shape[-3:]) - 1) // 2)
    slope = (k_max - k_min) / 2
    k = slope * nodes + ((k_max - k_min) / 2 + k_min)
    la = tke_spectrum(u_a - u_mean, k, p, w).log()
    lb = tke_spectrum(u_b - u_mean, k, p, w).log()
    D = torch.sqrt(slope * torch.einsum("ijk, k -> ij", (la[:, None] - lb[None]) ** 2, weights))
    return D, la, lb, k
