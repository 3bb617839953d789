"""An in-memory stand-in for the small part of the h5py API that the case files' reader uses (test infrastructure):
``File(path, mode)`` as a context manager, ``group["a/b"]``, ``.keys()``, ``.items()``, ``.attrs``, datasets that convert
with ``np.array(ds)`` and take sorted unique index arrays (h5py's fancy-indexing rule is enforced).  ``make_case`` builds
the tree of one case file as scripts/foam2h5.py + scripts/grid-embedding.py write it (seeded, so the golden generator
and the tests see the same bytes)."""

from pathlib import Path

import numpy as np

TREES = {}  # path (str) -> Group


class Dataset:
    def __init__(self, data, attrs=None):
        self.data = np.asarray(data)
        self.attrs = dict(attrs or {})

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def __getitem__(self, idx):
        if isinstance(idx, np.ndarray) and idx.ndim == 1:
            assert np.all(np.diff(idx) > 0), "h5py needs increasing, unique indices"
        return self.data[idx]

    @property
    def shape(self):
        return self.data.shape

    # -- the write side used by models.metrics.SampleStore.save_h5 (h5py.Dataset.resize / __setitem__)
    def resize(self, size, axis=0):
        assert self.maxshape is not None and self.maxshape[axis] is None, "only a dataset created resizable can grow"
        new = list(self.data.shape)
        new[axis] = size
        grown = np.zeros(new, dtype=self.data.dtype)
        sl = tuple(slice(0, min(a, b)) for a, b in zip(self.data.shape, new))
        grown[sl] = self.data[sl]
        self.data = grown

    def __setitem__(self, idx, value):
        self.data[idx] = value

    maxshape = None
    chunks = None


class Group:
    def __init__(self):
        self.children, self.attrs = {}, {}

    def __getitem__(self, path):
        node = self
        for part in str(path).split("/"):
            node = node.children[part]
        return node

    def keys(self):
        return self.children.keys()

    def items(self):
        return self.children.items()

    def put(self, path, value, attrs=None):
        parts = path.split("/")
        node = self
        for part in parts[:-1]:
            node = node.children.setdefault(part, Group())
        node.children[parts[-1]] = value if isinstance(value, Group) else Dataset(value, attrs)
        return node.children[parts[-1]]

    def group(self, path):
        node = self
        for part in path.split("/"):
            node = node.children.setdefault(part, Group())
        return node

    # -- the write side of h5py.Group
    require_group = group

    def __contains__(self, name):
        return name in self.children

    def create_dataset(self, name, data=None, chunks=None, maxshape=None):
        assert name not in self.children
        ds = self.children[name] = Dataset(np.array(data, copy=True))
        ds.chunks, ds.maxshape = chunks, maxshape
        return ds


class File:
    def __init__(self, path, mode="r"):
        if mode in ("a", "w") and (mode == "w" or str(path) not in TREES):
            TREES[str(path)] = Group()
        self.root = TREES[str(path)]

    def __enter__(self):
        return self.root

    def __exit__(self, *exc):
        return False


def make_case(seed, counts=(9, 7, 6), n_times=11):
    """One case file: data/{times,u,p,k,nut}, geometry/*, grid/{cell_counts,cell_idx,boundaries/*}, physical,
    boundary-conditions/<var>/<boundary> (fixed-value with a value dataset, zero-gradient, inlet-outlet)."""
    rng = np.random.default_rng(seed)
    root = Group()
    X, Y, Z = counts
    inside = np.zeros(counts, dtype=bool)
    inside[1:-1, 1:-1, 1:-1] = True
    inside[3:5, 2:4, 1:3] = False
    cell_idx = rng.permutation(np.flatnonzero(inside.ravel())).astype(np.int64)  # unsorted, as grid-embedding.py may leave it
    n = len(cell_idx)
    flat = np.arange(X * Y * Z).reshape(counts)
    root.put("data/times", np.round(np.linspace(0.0, 0.5, n_times) + seed * 1e-3, 4))
    root.put("data/u", rng.standard_normal((n_times, n, 3)).astype(np.float32))
    root.put("data/p", rng.standard_normal((n_times, n)).astype(np.float32))
    root.put("data/k", rng.standard_normal((n_times, n)).astype(np.float32))
    root.put("data/nut", rng.standard_normal((n_times, n)).astype(np.float32))
    root.put("geometry/bounding_box", np.array([0.4, 0.1, 0.1]) * (1 + seed))
    root.put("geometry/cell_counts", np.array([X - 2, Y - 2, Z - 2]))
    root.put("geometry/holes/positions", rng.random((2, 3)))
    root.put("geometry/holes/sizes", rng.random((2, 3)) * 0.1)
    root.put("grid/cell_counts", np.array(counts))
    root.put("grid/cell_idx", cell_idx)
    root.put("grid/boundaries/inlets", flat[0].ravel(), {"type": "patch", "start": 0, "n": Y * Z})
    root.put("grid/boundaries/outlets", flat[-1].ravel(), {"type": "patch", "start": 10, "n": Y * Z})
    root.put("grid/boundaries/walls", np.concatenate((flat[1:-1, 0].ravel(), flat[1:-1, -1].ravel())), {"type": "wall", "start": 20, "n": 2})
    root.group("physical").attrs["nu"] = 1e-4 * (seed + 1)
    bc = {"u": {"inlets": ("fixed-value", np.array([1.0 + seed, 0.0, 0.0], dtype=np.float32)), "walls": ("fixed-value", np.zeros(3, dtype=np.float32)),
                "outlets": ("inlet-outlet", None)},
          "p": {"inlets": ("zero-gradient", None), "walls": ("zero-gradient", None), "outlets": ("fixed-value", np.float32(0.0))}}
    for var, per in bc.items():
        for bname, (kind, value) in per.items():
            g = root.group(f"boundary-conditions/{var}/{bname}")
            g.attrs["type"] = kind
            if value is not None:
                g.put("value", value)
    return root


def install_cases(root_dir="/fake", phases=("train", "val"), per_phase=2):
    """Register seeded case trees under <root_dir>/<phase>/case-XX/data.h5; returns {phase: [paths]}."""
    out, seed = {}, 0
    for phase in phases:
        out[phase] = []
        for i in range(per_phase):
            path = Path(root_dir) / phase / f"case-{i:02d}" / "data.h5"
            TREES[str(path)] = make_case(seed, counts=(9, 7, 6) if seed % 2 == 0 else (8, 6, 7), n_times=11 + seed)
            out[phase].append(path)
            seed += 1
    return out


def dump_to_h5py(tree: Group, path):
    """Write a fake tree as a REAL HDF5 file (needs h5py: tests that call this importorskip it): the same groups,
    datasets and attributes, so the reader can be run over real files wherever h5py exists."""
    import h5py

    def rec(src, dst):
        for k, v in src.attrs.items():
            dst.attrs[k] = v
        for name, child in src.children.items():
            if isinstance(child, Group):
                rec(child, dst.require_group(name))
            else:
                ds = dst.create_dataset(name, data=child.data)
                for k, v in child.attrs.items():
                    ds.attrs[k] = v

    with h5py.File(path, "w") as f:
        rec(tree, f)
