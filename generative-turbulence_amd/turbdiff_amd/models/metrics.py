"""Sample metrics on the device (interface of turbdiff/models/metrics.py:220-380):
``interp3``, ``TurbulentKineticEnergySpectrum``, ``LogTKESpectrumL2Distance``, and ``SampleStore``
(metrics.py:36-124) kept in memory instead of an HDF5 file (h5py is not part of this package's requirements).

The spectrum's FFT is rocFFT (``torch.fft.fftn``); everything around it is two HIP kernels
(``tdx_tke_energy``, ``tdx_tke_sphere``: csrc/tdx_metrics.hip).  The Lebedev rule comes from
``scipy.integrate.lebedev_rule`` (the same 5810 nodes and weights as the reference's ``numgrids.pickle`` up to
their order; a reference checkpoint overwrites the ``p`` / ``w`` buffers with its own copy).  The Wasserstein
part of the reference's metrics (POT's EMD) stays where it is: on the CPU, outside this package.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from .. import _lib as L

# number of Lebedev nodes -> algebraic degree of the rule
_LEBEDEV_DEGREE = dict(zip(
    [6, 14, 26, 38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590, 770, 974, 1202, 1454, 1730, 2030, 2354,
     2702, 3074, 3470, 3890, 4334, 4802, 5294, 5810],
    [3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 35, 41, 47, 53, 59, 65, 71, 77, 83, 89, 95, 101, 107, 113, 119,
     125, 131]))


def lebedev_rule(n: int):
    """(points (n, 3), weights (n,) summing to 1) as float32 tensors."""
    from scipy.integrate import lebedev_rule as rule

    if n not in _LEBEDEV_DEGREE:
        raise RuntimeError(f"n={n} is not supported by numgrid.")
    x, w = rule(_LEBEDEV_DEGREE[n])
    return torch.tensor(x.T.copy()).float(), torch.tensor(w / (4 * math.pi)).float()


def interp3(grid: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """Trilinear interpolation of (..., X, Y, Z) grids at (N, 3) points -> (..., N) (metrics.py:220-268);
    plain torch indexing -- the spectrum below does not go through it."""
    hi = torch.tensor(grid.shape[-3:], dtype=torch.long, device=grid.device) - 1
    fl = torch.floor(points).long()
    p0 = torch.minimum(fl.clamp_min(0), hi)
    p1 = torch.minimum((fl + 1).clamp_min(0), hi)
    (x0, y0, z0), (x1, y1, z1) = p0.unbind(-1), p1.unbind(-1)
    wx, wy, wz = (points - p0).unbind(-1)
    g = grid
    return ((1 - wx) * (1 - wy) * (1 - wz) * g[..., x0, y0, z0] + (1 - wx) * (1 - wy) * wz * g[..., x0, y0, z1]
            + (1 - wx) * wy * (1 - wz) * g[..., x0, y1, z0] + (1 - wx) * wy * wz * g[..., x0, y1, z1]
            + wx * (1 - wy) * (1 - wz) * g[..., x1, y0, z0] + wx * (1 - wy) * wz * g[..., x1, y0, z1]
            + wx * wy * (1 - wz) * g[..., x1, y1, z0] + wx * wy * wz * g[..., x1, y1, z1])


class TurbulentKineticEnergySpectrum(nn.Module):
    """Estimate the turbulent kinetic energy spectrum of a 3D flow field (metrics.py:271-316)."""

    def __init__(self, n: int = 5810):
        super().__init__()
        self.n = n
        p, w = lebedev_rule(n)
        self.register_buffer("p", p)
        self.register_buffer("w", w)

    def forward(self, u_perturbation: torch.Tensor, k: torch.Tensor):
        u = u_perturbation
        if u.ndim < 4 or u.shape[-4] != 3 or u.dtype != torch.float32:
            raise RuntimeError(f"expected fp32 (..., 3, X, Y, Z), got {u.dtype} {tuple(u.shape)}")
        lead, (X, Y, Z) = u.shape[:-4], u.shape[-3:]
        u = u.reshape(-1, 3, X, Y, Z).contiguous()
        B, V = u.shape[0], X * Y * Z
        tke = torch.empty((B, X, Y, Z), dtype=torch.float32, device=u.device)
        L.call("tdx_tke_energy", L.ptr(u), L.ptr(tke), B, V, L.stream())
        spec = torch.view_as_real(torch.fft.fftn(tke, dim=(-3, -2, -1))).contiguous()  # rocFFT, unshifted
        k = k.to(u.device, torch.float32).contiguous()
        E = torch.empty((B, k.numel()), dtype=torch.float32, device=u.device)
        L.call("tdx_tke_sphere", L.ptr(spec), L.ptr(self.p.contiguous()), L.ptr(self.w.contiguous()), L.ptr(k), L.ptr(E), B,
               X, Y, Z, self.p.shape[0], k.numel(), L.stream())
        return E.reshape(*lead, k.numel())


class LogTKESpectrumL2Distance(nn.Module):
    """L2 distance between the log-TKE spectra of two flows by Gauss-Legendre integration (metrics.py:319-380)."""

    def __init__(self, tke_spectrum: nn.Module, n: int = 64):
        super().__init__()
        from scipy.special import roots_legendre

        self.tke_spectrum = tke_spectrum
        self.n = n
        nodes, weights = roots_legendre(n)
        self.register_buffer("legendre_nodes", torch.tensor(nodes).float())
        self.register_buffer("legendre_weights", torch.tensor(weights).float())

    def forward(self, u_a: torch.Tensor, u_b: torch.Tensor, u_mean: torch.Tensor):
        assert u_a.shape[-4] == 3 and u_b.shape[-4] == 3 and u_mean.shape[-4] == 3
        assert u_a.shape[-3:] == u_b.shape[-3:] == u_mean.shape[-3:]
        k_min, k_max = 1.0, float((min(u_a.shape[-3:]) - 1) // 2)
        slope = (k_max - k_min) / 2
        k = slope * self.legendre_nodes + ((k_max - k_min) / 2 + k_min)
        log_tke_a = self.tke_spectrum(u_a - u_mean, k).log()
        log_tke_b = self.tke_spectrum(u_b - u_mean, k).log()
        D = slope * torch.einsum("ijk, k -> ij", (log_tke_a[:, None] - log_tke_b[None]) ** 2, self.legendre_weights)
        return torch.sqrt(D), log_tke_a, log_tke_b, k


class SampleStore:
    """Generated samples per case, as the reference's ``SampleStore`` keeps them (metrics.py:36-124): for every
    variable the in-domain values, channels-last, ``(n_samples, n_cells, dims)`` -- in host memory instead of an
    HDF5 file.  ``samples_file`` is optional: ``save()`` writes one ``.npz`` with the arrays
    ``<case>/<variable>`` (the reference's ``<case>/data/<variable>`` datasets).

    Like the reference's, a store belongs to one process: under data-parallel evaluation every rank keeps the
    cases it sampled (``data.ofles.OpenFOAMEvaluationSampler`` shards whole batches) and rank 0 merges."""

    def __init__(self, samples_file=None, variables=()):
        self.samples_file = samples_file
        self.variables = tuple(variables)
        self._cases: dict = {}   # case name -> {variable: [tensor (b, n_cells, dims), ...]}

    # -- adding
    def add_cells(self, cells: dict, metadata):
        """``cells``: {variable: (B, n_cells, dims)} as ``Normalization.denormalized_cells`` / ``tdx_grid_select``
        produce them on the device (the gather + channels-last split of metrics.py:52-58 already done)."""
        case = self._cases.setdefault(metadata.case_name, {v: [] for v in self.variables})
        for v in self.variables:
            case[v].append(cells[v].detach().to("cpu", copy=True))

    def add_samples(self, x: torch.Tensor, metadata):
        """``x``: dense denormalised samples (B, F, X, Y, Z), as ``task.sample`` returns them (metrics.py:50-88)."""
        from ..data.ofles import split_channels
        from .utils import select_cells

        cells = select_cells(x, metadata.cell_idx.to(x.device)).transpose(-1, -2)  # (B, n_cells, F)
        self.add_cells(split_channels(cells, self.variables, dim=-1), metadata)

    # -- reading
    @property
    def case_names(self):
        return list(self._cases)

    def n_samples(self, case_name: str) -> int:
        return sum(t.shape[0] for t in self._cases[case_name][self.variables[0]])

    def load_samples(self, metadata, *, range=None):
        """All samples of the case described by ``metadata`` as an ``OpenFOAMData`` (metrics.py:95-112)."""
        from ..data.ofles import OpenFOAMData

        out = {}
        for v, parts in self._cases[metadata.case_name].items():
            t = torch.cat(parts) if parts else torch.empty(0)
            if range is not None:
                t = t[range]
                if t.ndim == 2:  # a single sample keeps its batch axis
                    t = t[None]
            out[v] = t
        return OpenFOAMData(metadata, torch.tensor([]), out)

    def reset(self):
        self._cases.clear()

    def save(self, path=None, opener=None):
        """``.h5`` / ``.hdf5``: the reference's samples file (``save_h5``); anything else: one ``.npz`` with the arrays
        ``<case>/<variable>``."""
        import numpy as np

        path = path or self.samples_file
        if str(path).endswith((".h5", ".hdf5")):
            return self.save_h5(path, opener=opener)
        arrays = {f"{case}/{v.name.lower()}": torch.cat(parts).numpy()
                  for case, per_var in self._cases.items() for v, parts in per_var.items() if parts}
        np.savez(path, **arrays)
        return path

    def save_h5(self, path=None, opener=None):
        """Append the stored samples to an HDF5 samples file laid out as the reference's ``SampleStore.add_samples``
        writes it (metrics.py:60-88): group ``<case>/data`` with one resizable dataset per variable, ``(n_samples, n_cells,
        dims)`` float32 chunked one sample per chunk, and the attribute ``n_samples``; a file that already holds
        samples of a case grows.  ``opener(path, mode)``: a context manager with h5py's File interface (default:
        ``h5py.File``, imported here -- the build image has none)."""
        from pathlib import Path

        path = path or self.samples_file
        op = opener or _open_h5py
        if opener is None:
            Path(path).parent.mkdir(parents=True, exist_ok=True)
        with op(path, "a") as f:
            for case, per_var in self._cases.items():
                data_group = f.require_group(case).require_group("data")
                n_prev = int(data_group.attrs.get("n_samples", 0))
                n_new = 0
                for v, parts in per_var.items():
                    if not parts:
                        continue
                    arr = torch.cat(parts).numpy()  # (n, n_cells, dims): scalars keep their unit axis, as torch.split leaves it
                    n_new = arr.shape[0]
                    name = v.name.lower()
                    if name not in data_group:
                        data_group.create_dataset(name, data=arr, chunks=arr[:1].shape, maxshape=(None, *arr.shape[1:]))
                    else:
                        ds = data_group[name]
                        if ds.shape[0] < n_prev + n_new:
                            ds.resize(n_prev + n_new, axis=0)
                        ds[n_prev : n_prev + n_new] = arr
                data_group.attrs["n_samples"] = n_prev + n_new
        return path

    @classmethod
    def from_h5(cls, path, variables, opener=None):
        """A store holding every case of a samples file written by the reference's store or by ``save_h5``."""
        import numpy as np

        store = cls(path, variables)
        with (opener or _open_h5py)(path, "r") as f:
            for case in f.keys():
                data_group = f[case]["data"]
                n = int(data_group.attrs.get("n_samples", -1))
                per = {}
                for v in store.variables:
                    t = torch.tensor(np.array(data_group[v.name.lower()]))
                    per[v] = [t[:n] if n >= 0 else t]
                store._cases[case] = per
        return store


def _open_h5py(path, mode="r"):
    try:
        import h5py
    except ImportError as e:
        raise ImportError("HDF5 sample files need h5py (pip install h5py), or pass opener= (an object with h5py's File "
                          "interface); SampleStore.save('x.npz') needs nothing") from e
    return h5py.File(path, mode)
