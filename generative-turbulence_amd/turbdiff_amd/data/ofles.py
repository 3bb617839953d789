"""In-memory data types of the reference's ``turbdiff/data/ofles.py`` that the hot path's callers touch
(SURVEY.md §8 f1): ``Variable``, ``BoundaryCondition``, ``OpenFOAMMetadata``, ``OpenFOAMData``
(``grid_embedding``), ``OpenFOAMStats`` (``normalizers``) and ``OpenFOAMBatch`` -- same names, fields and
semantics, without h5py / Lightning (reading ``data.h5`` and the samplers are row f2, not built here).

Everything that computes goes through ``turbdiff_amd.gridio`` (HIP kernels); the functions there only
rely on attribute names, so the reference's own objects can be passed as well.
"""

from __future__ import annotations

import pickle
from collections import defaultdict
from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path

import numpy as np
import torch


class Variable(Enum):
    """ofles.py:25-47."""

    U = 0
    P = 1
    K = 2
    NUT = 3
    CURL = 10
    ENSTROPHY = 11
    DIVERGENCE = 12
    GRAD = 13

    @property
    def dims(self) -> int:
        return {Variable.U: 3, Variable.CURL: 3, Variable.GRAD: 9}.get(self, 1)

    @staticmethod
    def from_str(name: str) -> "Variable":
        for v in Variable:
            if v.name.lower() == name.lower():
                return v
        raise RuntimeError(f"Unknown variable {name}")


@dataclass
class BoundaryCondition:
    """ofles.py:58-84 (``from_h5`` belongs to the HDF5 reader, row f2)."""

    class Type(Enum):
        FIXED_VALUE = 0
        ZERO_GRADIENT = 1
        INLET_OUTLET = 2

    type: "BoundaryCondition.Type"
    value: torch.Tensor | None = None


def split_channels(x: torch.Tensor, variables, *, dim=-4):
    """ofles.py:87-97."""
    return dict(zip(variables, torch.split(x, [v.dims for v in variables], dim=dim)))


@dataclass
class OpenFOAMMetadata:
    """ofles.py:106-193: the fields the path's callers read."""

    cell_counts: np.ndarray
    cell_idx: torch.Tensor
    boundaries: dict
    boundary_conditions: dict
    file: Path = Path("case/data.h5")
    nu: float = 0.0
    h: np.ndarray | None = None
    holes: list = field(default_factory=list)

    @property
    def device(self):
        return self.cell_idx.device

    @property
    def n_cells(self):
        return len(self.cell_idx)

    @property
    def case_name(self):
        return Path(self.file).parent.name

    @property
    def inside_mask(self):
        mask = torch.zeros(tuple(int(c) for c in self.cell_counts), device=self.device, dtype=torch.bool)
        mask.flatten()[self.cell_idx] = True
        return mask

    def to(self, device):
        b = {k: {**d, "idx": d["idx"].to(device)} for k, d in self.boundaries.items()}
        return OpenFOAMMetadata(self.cell_counts, self.cell_idx.to(device), b, self.boundary_conditions, self.file,
                                self.nu, self.h, self.holes)


@dataclass
class OpenFOAMData:
    """ofles.py:195-240."""

    metadata: OpenFOAMMetadata
    t: torch.Tensor
    samples: dict

    def __getattr__(self, name):
        if "metadata" in self.__dict__:
            return getattr(self.metadata, name)
        raise AttributeError(name)

    @property
    def n_samples(self):
        return next(iter(self.samples.values())).shape[0]

    @property
    def device(self):
        return self.metadata.cell_idx.device

    @property
    def variables(self):
        return tuple(self.samples.keys())

    def grid_embedding(self, variables):
        """(B, sum dims, X, Y, Z) fp32: zeros, the samples at ``cell_idx``, FIXED_VALUE boundary values."""
        from .. import gridio

        return gridio.grid_embed(self, variables)


@dataclass
class OpenFOAMStats:
    """ofles.py:243-303."""

    stats: dict
    _normalizers: dict = field(default_factory=dict)

    def normalizers(self, variables, mode: str):
        key = (tuple(variables), mode)
        if key in self._normalizers:  # the reference caches with cachedmethod
            return self._normalizers[key]
        if ":" in mode:
            mode_of = {Variable.from_str((pair := cfg.split(":"))[0]).name: pair[1] for cfg in mode.split(";")}
        else:
            orig = mode
            mode_of = defaultdict(lambda: orig)
        any_tensor = self.stats[variables[0].name.lower()]["mean"]
        dims = [v.dims for v in variables]
        mean, std = any_tensor.new_zeros(sum(dims)), any_tensor.new_ones(sum(dims))
        for v, mean_v, std_v in zip(variables, torch.split(mean, dims), torch.split(std, dims)):
            v_mode = mode_of[v.name]
            if "norm" in v_mode:
                st = self.stats[f"norm({v.name.lower()})"]
                if v_mode == "norm":
                    std_v[:] = st["mean"]
                elif v_mode == "norm-std":
                    mean_v[:] = st["mean"]
                    std_v[:] = st["std"]
                elif v_mode == "norm-max":
                    std_v[:] = st["max"]
                else:
                    raise RuntimeError(f"Unknown normalization mode {v_mode}")
            else:
                st = self.stats[v.name.lower()]
                if v_mode == "abs-max":
                    std_v[:] = torch.maximum(st["min"].abs(), st["max"].abs())
                elif v_mode == "mean-std":
                    mean_v[:] = st["mean"]
                    std_v[:] = st["std"]
                elif v_mode == "std":
                    std_v[:] = st["std"]
                else:
                    raise RuntimeError(f"Unknown normalization mode {v_mode}")
        std = torch.where(std >= 1e-8, std, 1.0)  # avoid division by 0 (ofles.py:291)
        self._normalizers[key] = (mean, std)
        return mean, std

    @staticmethod
    def from_file(file: Path):
        raw = pickle.loads(Path(file).read_bytes())
        return OpenFOAMStats({v: {n: torch.tensor(val) for n, val in st.items()} for v, st in raw.items()})

    def to(self, device):
        return OpenFOAMStats({v: {n: t.to(device) for n, t in st.items()} for v, st in self.stats.items()})


@dataclass
class OpenFOAMBatch:
    """ofles.py:306-309."""

    data: OpenFOAMData
    stats: OpenFOAMStats
