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

    _H5_TYPES = {"fixed-value": "FIXED_VALUE", "zero-gradient": "ZERO_GRADIENT", "inlet-outlet": "INLET_OUTLET"}

    @staticmethod
    def from_h5(group) -> "BoundaryCondition":
        """A ``boundary-conditions/<variable>/<boundary>`` group of a case file (ofles.py:68-84): attribute ``type``,
        and a ``value`` dataset for fixed values (a scalar for scalar fields)."""
        kind = group.attrs["type"]
        kind = kind.decode() if isinstance(kind, bytes) else kind
        name = BoundaryCondition._H5_TYPES.get(kind)
        if name is None:
            raise RuntimeError(f"Unknown boundary condition {group}")
        t = BoundaryCondition.Type[name]
        return BoundaryCondition(t, torch.tensor(np.array(group["value"])) if t is BoundaryCondition.Type.FIXED_VALUE else None)


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
# Repository, dataset, samplers and data module (ofles.py:312-643).  A repository is any object with `times` (one array
# of sample times per case file), `n_cases` and `read(file_idx, steps) -> OpenFOAMData`: OpenFOAMDataRepository reads the
# data.h5 case files through h5py's File interface (h5py itself is imported on first use: the build image has none, the
# tests drive the reader through an in-memory stand-in and pin it against the reference's reader run on the same
# stand-in), InMemoryRepository holds cases in memory.
import math
import random


def _chunked(seq, n):
    return [seq[i:i + n] for i in range(0, len(seq), n)]


@dataclass
class ChannelHole:
    """ofles.py:100-103."""

    pos: np.ndarray
    size: np.ndarray


def _open_h5(path, mode="r"):
    try:
        import h5py
    except ImportError as e:  # the build image has no h5py; the reader itself only needs its File / Group interface
        raise ImportError("reading data.h5 case files needs h5py (pip install h5py), or pass opener= to "
                          "OpenFOAMDataRepository") from e
    return h5py.File(path, mode)


class OpenFOAMDataRepository:
    """The case files' reader (reference ofles.py:320-421): ``times`` per file, ``read(file_idx, samples)`` ->
    ``OpenFOAMData``.  File layout (scripts/foam2h5.py:165-191, scripts/grid-embedding.py:74-90):
    ``data/{times, u, p, k, nut}`` with fields (T, n_cells[, 3]) fp32, ``grid/{cell_counts, cell_idx, boundaries/<name>}``
    (+ attribute ``type``), ``geometry/{bounding_box, cell_counts, holes/{positions, sizes}}``, ``physical`` (attribute
    ``nu``), ``boundary-conditions/<variable>/<boundary>``.  ``opener(path, mode)`` returns a context manager yielding
    an object with h5py's File interface (default: ``h5py.File``)."""

    def __init__(self, files, variables, opener=None):
        self.files, self.variables = list(files), tuple(variables)
        self._open = opener or _open_h5
        self.reset_caches()

    def reset_caches(self):
        self._metadata, self._times = {}, None

    @property
    def n_cases(self):
        return len(self.files)

    @property
    def times(self):
        if self._times is None:
            self._times = []
            for path in self.files:
                with self._open(path, "r") as f:
                    self._times.append(np.array(f["data/times"]).copy())
        return self._times

    def read_metadata(self, file_idx: int) -> OpenFOAMMetadata:
        meta = self._metadata.get(file_idx)  # geometry is read once per file (the reference memoises the method)
        if meta is not None:
            return meta
        with self._open(self.files[file_idx], "r") as f:
            geo, grid = f["geometry"], f["grid"]
            h = torch.tensor(np.array(geo["bounding_box"])) / torch.tensor(np.array(geo["cell_counts"]))
            positions, sizes = np.array(geo["holes/positions"]).copy(), np.array(geo["holes/sizes"]).copy()
            boundaries = {}
            for name in grid["boundaries"].keys():
                ds = grid["boundaries"][name]
                boundaries[name] = {"type": ds.attrs["type"], "idx": torch.tensor(np.array(ds))}
            conditions = {Variable.from_str(var): {b: BoundaryCondition.from_h5(g) for b, g in per_boundary.items()}
                          for var, per_boundary in f["boundary-conditions"].items()}
            meta = OpenFOAMMetadata(cell_counts=np.array(grid["cell_counts"]).copy(), cell_idx=torch.tensor(np.array(grid["cell_idx"])),
                                    boundaries=boundaries, boundary_conditions=conditions, file=self.files[file_idx],
                                    nu=f["physical"].attrs["nu"], h=h,
                                    holes=[ChannelHole(p, s) for p, s in zip(positions, sizes)])
        self._metadata[file_idx] = meta
        return meta

    def read_data(self, file_idx: int, sample_idxs) -> dict:
        """{variable: (len(sample_idxs), n_cells, dims)}: HDF5 wants increasing unique row indices, so the rows are
        read once in sorted order and spread back to the request's order (duplicates included)."""
        want = np.asarray(sample_idxs)
        rows, back = np.unique(want, return_inverse=True)
        out = {}
        with self._open(self.files[file_idx], "r") as f:
            fields = f["data"]
            for v in self.variables:
                block = torch.tensor(np.array(fields[v.name.lower()][rows]))
                if block.ndim == 2:  # scalar field: (T, n_cells) -> (T, n_cells, 1)
                    block = block.unsqueeze(-1)
                out[v] = block[back]
        return out

    def read(self, file_idx: int, samples) -> OpenFOAMData:
        return OpenFOAMData(self.read_metadata(file_idx), torch.tensor(self.times[file_idx][samples]), self.read_data(file_idx, samples))


def find_data_files(cases_root: Path, exists=None):
    """``<cases_root>/<case>/data.h5`` of every case directory, without walking the case directories (ofles.py:548-551)."""
    exists = exists or (lambda p: p.is_file())
    return [p for d in sorted(Path(cases_root).iterdir()) if exists(p := d / "data.h5")]


def reset_dataset_caches(worker_id):
    info = torch.utils.data.get_worker_info()
    if info is not None:
        info.dataset.reset_caches()


class OpenFOAMDataModule:
    """ofles.py:564-643 without Lightning: ``setup(stage)`` loads ``<root>/stats.pickle`` and builds the datasets of
    ``<root>/{train,val,test}``; the three loaders are ``DataLoader(dataset, sampler=..., batch_size=None)`` with the
    reference's samplers.  ``rank`` / ``world_size`` / ``seed`` shard the training batches for data-parallel runs (every
    rank the same number of whole single-geometry batches)."""

    def __init__(self, root: Path, discard_first_seconds: float, num_workers: int = 2, batch_size: int = 1,
                 eval_batch_size: int = 8, val_samples: int = 8, test_samples: int = 32, pin_memory: bool = True,
                 variables=tuple(Variable), *, rank: int = 0, world_size: int = 1, seed=None, opener=None, list_cases=None):
        self.root, self.discard_first_seconds, self.num_workers = Path(root), discard_first_seconds, num_workers
        self.batch_size, self.eval_batch_size = batch_size, eval_batch_size
        self.val_samples, self.test_samples, self.pin_memory, self.variables = val_samples, test_samples, pin_memory, tuple(variables)
        self.rank, self.world_size, self.seed = rank, world_size, seed
        self._opener, self._list_cases = opener, list_cases or find_data_files
        self.train_dataset = self.val_dataset = self.test_dataset = None
        self.stats = None

    def setup(self, stage: str, stats=None):
        self.stats = stats if stats is not None else OpenFOAMStats.from_file(self.root / "stats.pickle")
        wanted = {"fit": ("train", "val"), "validate": ("val",), "test": ("test",)}.get(stage, ())
        for phase in wanted:
            if getattr(self, f"{phase}_dataset") is None:
                repo = OpenFOAMDataRepository(self._list_cases(self.root / phase), self.variables, opener=self._opener)
                setattr(self, f"{phase}_dataset", OpenFOAMDataset(repo, self.stats, self.discard_first_seconds))

    def _loader(self, dataset, sampler):
        return torch.utils.data.DataLoader(dataset, sampler=sampler, worker_init_fn=reset_dataset_caches, batch_size=None,
                                           num_workers=self.num_workers, pin_memory=self.pin_memory)

    def train_dataloader(self):
        return self._loader(self.train_dataset, OpenFOAMSampler(self.train_dataset, batch_size=self.batch_size, shuffle=True,
                                                                rank=self.rank, world_size=self.world_size, seed=self.seed))

    def val_dataloader(self):
        return self._loader(self.val_dataset, OpenFOAMEvaluationSampler(self.val_dataset, batch_size=self.eval_batch_size,
                                                                        samples_per_file=self.val_samples))

    def test_dataloader(self):
        return self._loader(self.test_dataset, OpenFOAMEvaluationSampler(self.test_dataset, batch_size=self.eval_batch_size,
                                                                         samples_per_file=self.test_samples))


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
