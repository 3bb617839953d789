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

    # normalisation rule of a mode: which statistics record it reads ("norm(<v>)" = statistics of the vector norm, or
    # the per-component record "<v>"), which entry becomes the shift (None: 0) and which the scale
    _RULES = {
        "norm": ("norm({})", None, lambda st: st["mean"]),
        "norm-std": ("norm({})", "mean", lambda st: st["std"]),
        "norm-max": ("norm({})", None, lambda st: st["max"]),
        "abs-max": ("{}", None, lambda st: torch.maximum(st["min"].abs(), st["max"].abs())),
        "mean-std": ("{}", "mean", lambda st: st["std"]),
        "std": ("{}", None, lambda st: st["std"]),
    }

    @staticmethod
    def _modes_by_variable(variables, mode: str) -> dict:
        """"u:norm-max;p:abs-max" -> one mode per variable; a plain mode name applies to every variable."""
        if ":" not in mode:
            return {v: mode for v in variables}
        given = dict(item.split(":") for item in mode.split(";"))
        by_var = {Variable.from_str(name): m for name, m in given.items()}
        return {v: by_var[v] for v in variables}  # KeyError: a variable without a mode, as in the reference

    def normalizers(self, variables, mode: str):
        """(shift, scale), each (sum of dims,): x_normalised = (x - shift) / scale  (reference ofles.py:248-293)."""
        key = (tuple(variables), mode)
        hit = self._normalizers.get(key)  # the reference memoises with cachedmethod
        if hit is not None:
            return hit
        shifts, scales = [], []
        for v, v_mode in self._modes_by_variable(variables, mode).items():
            rule = self._RULES.get(v_mode)
            if rule is None:
                raise RuntimeError(f"Unknown normalization mode {v_mode}")
            record, shift_key, scale_of = rule
            st = self.stats[record.format(v.name.lower())]
            scale = scale_of(st)
            shift = st[shift_key] if shift_key is not None else torch.zeros_like(scale)
            # scalar statistics (those of a vector norm) apply to every component of the variable
            shifts.append(shift.reshape(-1).expand(v.dims) if shift.numel() == 1 else shift.reshape(v.dims))
            scales.append(scale.reshape(-1).expand(v.dims) if scale.numel() == 1 else scale.reshape(v.dims))
        shift, scale = torch.cat(shifts), torch.cat(scales)
        scale = torch.where(scale >= 1e-8, scale, torch.ones_like(scale))  # no division by ~0 (ofles.py:291)
        self._normalizers[key] = (shift, scale)
        return shift, scale

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


# ---------------------------------------------------------------------------------------------------------
# Dataset and samplers (ofles.py:424-540).  The repository is any object with `times` (one array of sample
# times per case file), `n_cases` and `read(file_idx, steps) -> OpenFOAMData`; the reference's HDF5-backed
# OpenFOAMDataRepository (ofles.py:312-421) satisfies it, and so does InMemoryRepository below.  h5py is not
# part of this package's requirements, so no file reader is provided.
import math
import random


def _chunked(seq, n):
    return [seq[i:i + n] for i in range(0, len(seq), n)]


class InMemoryRepository:
    """Cases held in memory: `cases[i]` = (OpenFOAMMetadata, times (T,), {Variable: (T, n_cells, dims) tensor})."""

    def __init__(self, cases):
        self.cases = cases
        self.times = [np.asarray(c[1]) for c in cases]

    @property
    def n_cases(self):
        return len(self.cases)

    def reset_caches(self):
        pass

    def read(self, file_idx: int, samples):
        meta, times, fields = self.cases[file_idx]
        idx = torch.as_tensor(np.asarray(samples), dtype=torch.long)
        return OpenFOAMData(meta, torch.as_tensor(np.asarray(times))[idx], {v: f[idx] for v, f in fields.items()})


class OpenFOAMDataset(torch.utils.data.Dataset):
    """ofles.py:424-479: flat sample index over all cases; a batch is a list of indices of ONE case."""

    def __init__(self, repo, stats, discard_first_seconds: float):
        super().__init__()
        self.repo, self.stats, self.discard_first_seconds = repo, stats, discard_first_seconds
        self.reset_caches()

    def reset_caches(self):
        self.repo.reset_caches()
        self.valid_steps = [np.nonzero(np.asarray(t) > self.discard_first_seconds)[0] for t in self.repo.times]
        # case c owns the flat sample indices [first[c], first[c + 1])
        self._first = np.concatenate(([0], np.cumsum([len(v) for v in self.valid_steps]))).astype(np.int64)
        self._step_of_time = {}

    def sample_idxs_by_file(self):
        return [list(range(int(a), int(b))) for a, b in zip(self._first[:-1], self._first[1:])]

    def __len__(self):
        return int(self._first[-1])

    def __getitem__(self, index):
        """A batch = flat indices that all fall into ONE case (same geometry), reference ofles.py:441-468."""
        flat = np.atleast_1d(np.asarray(index, dtype=np.int64))
        case = int(np.searchsorted(self._first, flat.min(), side="right")) - 1
        assert flat.max() < self._first[case + 1], "All samples have to be from the same geometry"
        steps = self.valid_steps[case][flat - self._first[case]]
        return OpenFOAMBatch(self.repo.read(case, list(steps)), self.stats)

    def get_times(self, file_idx: int, times):
        """The samples of a case at the given physical times, matched at 0.1 ms resolution (ofles.py:470-479)."""
        lookup = self._step_of_time.get(file_idx)
        if lookup is None:
            lookup = {}
            for step, t in enumerate(np.asarray(self.repo.times[file_idx])):
                lookup.setdefault(int(round(float(t) * 10_000)), step)  # first occurrence wins, like list.index
            self._step_of_time[file_idx] = lookup
        try:
            steps = [lookup[int(round(float(t) * 10_000))] for t in times]
        except KeyError as e:
            raise ValueError(f"no sample at time key {e.args[0]} in case {file_idx}") from None
        return OpenFOAMBatch(self.repo.read(file_idx, steps), self.stats)


class OpenFOAMSampler(torch.utils.data.Sampler):
    """ofles.py:482-511, plus data-parallel sharding (shard at batch-list granularity).

    With ``world_size == 1`` and ``seed is None`` it is the reference's sampler, draw for draw (it uses the
    global ``random`` state).  With ``seed`` set, epoch e shuffles with ``random.Random(seed + e)`` -- the same
    order on every rank -- and rank r takes batches r, r + W, ... of it; the list is extended by wrapping
    around so that every rank gets the same number of batches (the gradient all-reduce needs lock-step)."""

    def __init__(self, dataset, *, batch_size: int, shuffle: bool, rank: int = 0, world_size: int = 1, seed=None):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        if world_size > 1 and shuffle and seed is None:
            raise ValueError("sharded shuffling needs a seed shared by all ranks")

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _n_total(self):
        return sum(math.ceil(len(s) / self.batch_size) for s in self.dataset.valid_steps)

    def __len__(self):
        return math.ceil(self._n_total() / self.world_size)

    def __iter__(self):
        rng = random if self.seed is None else random.Random(self.seed + self.epoch)
        indices = self.dataset.sample_idxs_by_file()
        if self.shuffle:
            for idxs in indices:
                rng.shuffle(idxs)
        batches = []
        for idxs in indices:
            batches.extend(_chunked(idxs, self.batch_size))
        if self.shuffle:
            rng.shuffle(batches)
        if self.world_size > 1:
            n = len(self) * self.world_size
            batches = (batches * math.ceil(n / len(batches)))[:n][self.rank::self.world_size]
        yield from batches


class OpenFOAMEvaluationSampler(torch.utils.data.Sampler):
    """ofles.py:514-545: evenly spaced samples of every case; optionally sharded (whole batches per rank,
    no padding: evaluation has no collective)."""

    def __init__(self, dataset, *, batch_size: int, samples_per_file: int, rank: int = 0, world_size: int = 1):
        self.dataset, self.batch_size, self.samples_per_file = dataset, batch_size, samples_per_file
        self.rank, self.world_size = rank, world_size

    def _batches(self):
        batches = []
        for idxs in self.dataset.sample_idxs_by_file():
            pick = np.round(np.linspace(0, len(idxs) - 1, num=self.samples_per_file)).astype(int)
            batches.extend(_chunked([idxs[i] for i in pick], self.batch_size))
        return batches

    def __len__(self):
        n = self.dataset.repo.n_cases * math.ceil(self.samples_per_file / self.batch_size)
        return len(range(self.rank, n, self.world_size))

    def __iter__(self):
        yield from self._batches()[self.rank::self.world_size]
