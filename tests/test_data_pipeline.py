"""Dataset / samplers / device staging (the part that does not need h5py): host logic against
golden vectors from the reference's own classes (tests/golden/make_golden_sampler.py), sharding properties,
and the staged batches on the GPU."""

import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
from turbdiff_amd.data.ofles import (InMemoryRepository, OpenFOAMBatch, OpenFOAMDataset, OpenFOAMEvaluationSampler,
                                     OpenFOAMMetadata, OpenFOAMSampler, OpenFOAMStats, Variable)


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN / "samplers.npz")


class FakeRepo:
    def __init__(self, g):
        self.times = [g[f"times/{i}"] for i in range(3)]
        self.n_cases = 3

    def reset_caches(self):
        pass

    def read(self, file_idx, samples):
        return ("case", file_idx, [int(s) for s in samples])


def unflat(sizes, idx):
    out, i = [], 0
    for n in sizes:
        out.append([int(v) for v in idx[i:i + n]])
        i += n
    return out


def test_dataset_matches_reference(g):
    ds = OpenFOAMDataset(FakeRepo(g), "STATS", float(g["discard"]))
    assert len(ds) == int(g["len"])
    for i in range(3):
        assert np.array_equal(ds.valid_steps[i], g[f"valid_steps/{i}"])
    for i, batch in enumerate(([0], [3, 1, 2], [29 + 2, 29 + 0], [29 + 7 + 5])):
        b = ds[list(batch)]
        assert b.stats == "STATS" and b.data[1] == int(g["getitem/files"][i]) and b.data[2] == list(g[f"getitem/steps/{i}"])
    assert ds.get_times(2, [0.1, 0.3]).data[2] == list(g["get_times/steps"])
    with pytest.raises(AssertionError, match="same geometry"):
        ds[[28, 29]]  # last sample of case 0 and first of case 1


@pytest.mark.parametrize("bs", [1, 4, 6])
def test_train_sampler_is_the_reference_sampler(g, bs):
    ds = OpenFOAMDataset(FakeRepo(g), None, float(g["discard"]))
    s = OpenFOAMSampler(ds, batch_size=bs, shuffle=False)
    assert len(s) == int(g[f"train/bs{bs}/len"])
    assert list(s) == unflat(g[f"train/bs{bs}/plain/sizes"], g[f"train/bs{bs}/plain/idx"])
    random.seed(1000 + bs)  # the reference shuffles with the global RNG
    s = OpenFOAMSampler(ds, batch_size=bs, shuffle=True)
    assert list(s) == unflat(g[f"train/bs{bs}/shuffled/sizes"], g[f"train/bs{bs}/shuffled/idx"])


@pytest.mark.parametrize("bs,spf", [(8, 8), (3, 5)])
def test_eval_sampler_is_the_reference_sampler(g, bs, spf):
    ds = OpenFOAMDataset(FakeRepo(g), None, float(g["discard"]))
    s = OpenFOAMEvaluationSampler(ds, batch_size=bs, samples_per_file=spf)
    assert len(s) == int(g[f"eval/bs{bs}_spf{spf}/len"])
    assert list(s) == unflat(g[f"eval/bs{bs}_spf{spf}/sizes"], g[f"eval/bs{bs}_spf{spf}/idx"])
    parts = [list(OpenFOAMEvaluationSampler(ds, batch_size=bs, samples_per_file=spf, rank=r, world_size=3)) for r in range(3)]
    assert sorted(map(tuple, sum(parts, []))) == sorted(map(tuple, s)) and [len(p) for p in parts] == [
        len(OpenFOAMEvaluationSampler(ds, batch_size=bs, samples_per_file=spf, rank=r, world_size=3)) for r in range(3)]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_train_sampler_properties(g, world):
    ds = OpenFOAMDataset(FakeRepo(g), None, float(g["discard"]))
    full = OpenFOAMSampler(ds, batch_size=4, shuffle=True, seed=11)
    full.set_epoch(3)
    ref = list(full)
    shards = []
    for r in range(world):
        s = OpenFOAMSampler(ds, batch_size=4, shuffle=True, rank=r, world_size=world, seed=11)
        s.set_epoch(3)
        shards.append(list(s))
        assert len(shards[-1]) == len(s)
    assert len({len(s) for s in shards}) == 1                       # lock-step: same number of batches per rank
    merged = [b for i in range(len(shards[0])) for b in (s[i] for s in shards)]
    assert merged[: len(ref)] == ref                                # the ranks interleave the single-process order
    assert all(b in ref for b in merged[len(ref):])                 # padding wraps around
    assert all(len({i >= 29 for i in b}) == 1 for b in merged)      # every batch stays inside one case
    other = OpenFOAMSampler(ds, batch_size=4, shuffle=True, seed=11)
    other.set_epoch(4)
    assert list(other) != ref                                       # epochs reshuffle
    with pytest.raises(ValueError):
        OpenFOAMSampler(ds, batch_size=4, shuffle=True, rank=0, world_size=2)


def _cases():
    gen = torch.Generator().manual_seed(0)
    cases = []
    for n, counts in ((5, (6, 5, 4)), (3, (5, 5, 5))):
        inside = torch.zeros(counts, dtype=torch.bool)
        inside[1:-1, 1:-1, 1:-1] = True
        cell_idx = inside.flatten().nonzero().flatten()
        meta = OpenFOAMMetadata(np.array(counts), cell_idx, {"walls": {"idx": torch.tensor([0, 1])}}, {})
        cases.append((meta, np.arange(n) * 0.1, {Variable.U: torch.randn(n, len(cell_idx), 3, generator=gen),
                                                 Variable.P: torch.randn(n, len(cell_idx), 1, generator=gen)}))
    return cases


def test_in_memory_repository_feeds_the_dataset():
    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3)}, "p": {"mean": torch.tensor(0.0), "std": torch.tensor(1.0)}})
    cases = _cases()
    ds = OpenFOAMDataset(InMemoryRepository(cases), stats, discard_first_seconds=0.05)
    assert len(ds) == 4 + 2
    b = ds[[1, 3]]
    assert isinstance(b, OpenFOAMBatch) and b.data.n_samples == 2 and b.data.metadata is cases[0][0]
    assert torch.equal(b.data.samples[Variable.U], cases[0][2][Variable.U][[2, 4]])
    assert torch.allclose(b.data.t, torch.tensor([0.2, 0.4], dtype=b.data.t.dtype))


@pytest.mark.gpu
def test_device_stager_delivers_the_batches_in_order():
    from turbdiff_amd.data.staging import DeviceStager

    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3)}, "p": {"mean": torch.tensor(0.0), "std": torch.tensor(1.0)}})
    ds = OpenFOAMDataset(InMemoryRepository(_cases()), stats, discard_first_seconds=-1.0)
    order = list(OpenFOAMSampler(ds, batch_size=2, shuffle=False)) * 3
    want = [ds[b] for b in order]
    got = list(DeviceStager((ds[b] for b in order), "cuda:0"))
    assert len(got) == len(want)
    metas = set()
    for a, b in zip(got, want):
        for v in (Variable.U, Variable.P):
            assert a.data.samples[v].is_cuda and torch.equal(a.data.samples[v].cpu(), b.data.samples[v])
        assert a.data.metadata.cell_idx.is_cuda and torch.equal(a.data.metadata.cell_idx.cpu(), b.data.metadata.cell_idx)
        metas.add(id(a.data.metadata))
        x = a.data.grid_embedding((Variable.U, Variable.P))  # the staged batch feeds the ingress kernel
        assert x.shape[0] == a.data.n_samples and torch.isfinite(x).all()
    assert len(metas) == 2  # each geometry moved to the device once


@pytest.mark.gpu
def test_trainer_runs_from_dataset_sampler_and_stager():
    """The whole caller chain on the device: repository -> dataset -> (sharded) sampler -> pinned staging ->
    DiffusionTrainer.fit_step (fused ingress, U-Net step, ClipRAdam)."""
    from turbdiff_amd.data.staging import DeviceStager
    from turbdiff_amd.training import DiffusionTrainer

    gen = torch.Generator().manual_seed(1)
    counts = (12, 10, 9)
    inside = torch.zeros(counts, dtype=torch.bool)
    inside[1:-1, 1:-1, 1:-1] = True
    cell_idx = inside.flatten().nonzero().flatten()
    idx = torch.arange(inside.numel()).reshape(counts)
    from turbdiff_amd.data.ofles import BoundaryCondition as BC

    meta = OpenFOAMMetadata(np.array(counts), cell_idx, {"walls": {"idx": idx[:, 0].flatten()}, "inlets": {"idx": idx[0].flatten()}},
                            {Variable.U: {"inlets": BC(BC.Type.FIXED_VALUE, torch.tensor([1.0, 0.0, 0.0]))}})
    fields = {Variable.U: torch.randn(8, len(cell_idx), 3, generator=gen), Variable.P: torch.randn(8, len(cell_idx), 1, generator=gen)}
    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3) * 1.5}, "p": {"mean": torch.tensor(0.1), "std": torch.tensor(0.8)}})
    ds = OpenFOAMDataset(InMemoryRepository([(meta, np.arange(8) * 0.1, fields)]), stats, discard_first_seconds=-1.0)
    sampler = OpenFOAMSampler(ds, batch_size=2, shuffle=True, rank=1, world_size=2, seed=5)
    torch.manual_seed(0)
    tr = DiffusionTrainer(**{**DiffusionTrainer.SHIPPED_CONFIG, "dim": 8, "timesteps": 10}, u_net_levels=2, normalization_mode="mean-std",
                          max_train_steps=10).to("cuda:0")
    losses = [tr.fit_step(b).item() for b in DeviceStager((ds[i] for i in sampler), "cuda:0")]
    assert len(losses) == len(sampler) == 2 and all(np.isfinite(losses))
    assert tr.cell_type_embedding.embedding.weight.grad is None or torch.isfinite(tr.cell_type_embedding.embedding.weight).all()


@pytest.mark.gpu
def test_eval_ckpt_flow_on_in_memory_cases(tmp_path):
    """The flow of the reference's scripts/eval_ckpt.py:43-76 (tools/eval_ckpt.py): run configuration from the
    checkpoint -> overrides -> task from the configuration -> strict load -> sample every validation batch ->
    SampleStore -> metrics; on synthetic cases held by InMemoryRepository, with a small model and a short reverse
    process.  Checks the plumbing: every case sampled, store layout as SampleStore.add_samples keeps it, the stored
    values equal a direct sample_cells() replay, reruns reproduce, and the CLI prints the metric."""
    import importlib.util
    import json
    import subprocess

    from turbdiff_amd.data.ofles import Variable
    from turbdiff_amd.training import DiffusionTrainer

    spec = importlib.util.spec_from_file_location("eval_ckpt", ROOT / "tools" / "eval_ckpt.py")
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    run = json.loads((ROOT / "tests" / "golden" / "task_configs.json").read_text())["shipped"]["run_config"]
    run["model"].update(dim=8, timesteps=10, eval_batch_size=2)
    run["matmul_precision"] = "highest"
    torch.manual_seed(3)
    src = DiffusionTrainer.from_config(run, max_train_steps=1)
    ckpt = {"config": run, "state_dict": {k: v.clone() for k, v in src.state_dict().items()}}
    cases, stats = ev.synthetic_cases(2, grid=(24, 16, 16), n_times=6)
    dev = torch.device("cuda:0")
    store, metrics, task = ev.evaluate(ckpt, cases, stats, dev, overrides=["model.noise_bcs=false"], val_samples=3, start_from=4,
                                       samples_path=tmp_path / "samples.npz")
    assert task.noise_bcs is False and task.compute_mode == "f32"
    for k, v in task.state_dict().items():
        assert torch.equal(v.cpu(), ckpt["state_dict"][k]), k
    assert store.case_names == [c[0].case_name for c in cases]
    for meta, _, _ in cases:
        data = store.load_samples(meta)
        assert data.samples[Variable.U].shape == (3, meta.n_cells, 3) and data.samples[Variable.P].shape == (3, meta.n_cells, 1)
        assert torch.isfinite(data.samples[Variable.U]).all()
    z = np.load(tmp_path / "samples.npz")
    assert sorted(z.files) == sorted(f"{c[0].case_name}/{v}" for c in cases for v in ("u", "p"))
    assert np.isfinite(list(metrics.values())).all() and "val/log_tke_l2" in metrics
    # same seed, same checkpoint -> the same samples
    store2, metrics2, _ = ev.evaluate(ckpt, cases, stats, dev, overrides=["model.noise_bcs=false"], val_samples=3, start_from=4)
    for meta, _, _ in cases:
        assert torch.equal(store2.load_samples(meta).samples[Variable.U], store.load_samples(meta).samples[Variable.U])
    assert metrics2 == metrics
    # the command line
    torch.save(ckpt, tmp_path / "model.ckpt")
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "eval_ckpt.py"), str(tmp_path / "model.ckpt"), str(tmp_path / "cli.npz"),
                          "--synthetic", "1", "--start-from", "2", "model.eval_batch_size=2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "val/log_tke_l2:" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


# --------------------------------------------------------------------------- the case-file reader (§8 f2)


def test_repository_reads_case_files_like_the_reference():
    """OpenFOAMDataRepository through h5py's File interface (tests/h5fake.py stands in for h5py, which the image
    lacks) against the reference's own reader run on the same trees (tests/golden/make_golden_repository.py): times,
    every metadata field, and reads of unsorted index lists with duplicates."""
    import h5fake
    from turbdiff_amd.data.ofles import OpenFOAMDataRepository

    z = np.load(GOLDEN / "repository.npz")
    files = h5fake.install_cases()
    requests = [[5, 2, 2, 7], [0], [10, 9, 8, 0, 10], [3, 4, 5]]
    for phase, paths in files.items():
        repo = OpenFOAMDataRepository(paths, (Variable.U, Variable.P, Variable.NUT), opener=h5fake.File)
        assert repo.n_cases == len(paths)
        for i in range(repo.n_cases):
            k = f"{phase}/{i}"
            assert np.array_equal(repo.times[i], z[f"{k}/times"])
            m = repo.read_metadata(i)
            assert repo.read_metadata(i) is m, "geometry must be read once per file"
            assert np.array_equal(m.cell_counts, z[f"{k}/cell_counts"]) and np.array_equal(m.cell_idx.numpy(), z[f"{k}/cell_idx"])
            assert np.array_equal(m.h.numpy(), z[f"{k}/h"]) and m.nu == float(z[f"{k}/nu"]) and m.case_name == str(z[f"{k}/case_name"])
            assert np.array_equal(np.stack([np.concatenate((h.pos, h.size)) for h in m.holes]), z[f"{k}/holes"])
            assert sorted(m.boundaries) == sorted(n.split("/")[3] for n in z.files if n.startswith(f"{k}/boundary/") and n.endswith("/idx"))
            for name, desc in m.boundaries.items():
                assert np.array_equal(desc["idx"].numpy(), z[f"{k}/boundary/{name}/idx"]) and desc["type"] == str(z[f"{k}/boundary/{name}/type"])
            for var, per in m.boundary_conditions.items():
                for bname, bc in per.items():
                    assert bc.type.name == str(z[f"{k}/bc/{var.name}/{bname}/type"])
                    if bc.value is not None:
                        assert np.array_equal(bc.value.numpy(), z[f"{k}/bc/{var.name}/{bname}/value"])
            for r, req in enumerate(requests):
                data = repo.read(i, req)
                assert np.array_equal(data.t.numpy(), z[f"{k}/read/{r}/t"])
                assert [v.name for v in data.samples] == ["U", "P", "NUT"]
                for v, x in data.samples.items():
                    assert x.dtype == torch.float32 and np.array_equal(x.numpy(), z[f"{k}/read/{r}/{v.name}"])


def test_data_module_builds_datasets_and_loaders():
    """OpenFOAMDataModule (ofles.py:564-643 without Lightning) on the stand-in case files: datasets per phase, the
    reference's samplers behind batch_size=None loaders, batches of one geometry each, sharded training batches."""
    import h5fake
    from turbdiff_amd.data.ofles import OpenFOAMDataModule

    files = h5fake.install_cases(root_dir="/fake2", phases=("train", "val", "test"), per_phase=2)
    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3)}, "p": {"mean": torch.tensor(0.0), "std": torch.tensor(1.0)}})
    listing = lambda d: files[Path(d).name]
    dm = OpenFOAMDataModule("/fake2", discard_first_seconds=0.1, num_workers=0, batch_size=3, eval_batch_size=2, val_samples=4,
                            test_samples=3, pin_memory=False, variables=(Variable.U, Variable.P), opener=h5fake.File, list_cases=listing)
    dm.setup("fit", stats=stats)
    assert dm.train_dataset is not None and dm.val_dataset is not None and dm.test_dataset is None
    random.seed(0)
    batches = list(dm.train_dataloader())
    assert len(batches) == len(dm.train_dataloader().sampler) and all(isinstance(b, OpenFOAMBatch) for b in batches)
    for b in batches:
        assert b.data.samples[Variable.U].shape[1:] == (b.data.metadata.n_cells, 3) and b.data.n_samples <= 3
        assert float(b.data.t.min()) > 0.1  # discard_first_seconds
    val = list(dm.val_dataloader())
    assert sum(b.data.n_samples for b in val) == 2 * 4 and all(b.data.n_samples <= 2 for b in val)
    dm.setup("test", stats=stats)
    assert sum(b.data.n_samples for b in dm.test_dataloader()) == 2 * 3
    # two ranks draw disjoint halves of one seeded shuffle, in lock step
    shards = []
    for r in range(2):
        d2 = OpenFOAMDataModule("/fake2", 0.1, num_workers=0, batch_size=3, pin_memory=False, variables=(Variable.U, Variable.P),
                                opener=h5fake.File, list_cases=listing, rank=r, world_size=2, seed=7)
        d2.setup("fit", stats=stats)
        shards.append([tuple(np.round(b.data.t.numpy(), 4)) + (b.data.metadata.case_name,) for b in d2.train_dataloader()])
    assert len(shards[0]) == len(shards[1]) and not set(shards[0]) & set(shards[1])


def _store_with_samples(seed=0):
    from turbdiff_amd.data.ofles import OpenFOAMMetadata, Variable
    from turbdiff_amd.models.metrics import SampleStore

    import numpy as np

    g = torch.Generator().manual_seed(seed)
    V = (Variable.U, Variable.P)
    store = SampleStore(None, V)
    metas = []
    for i, n_cells in enumerate((37, 50)):
        md = OpenFOAMMetadata(cell_counts=np.array([6, 5, 4]), cell_idx=torch.arange(n_cells), boundaries={}, boundary_conditions={},
                              file=Path(f"/x/case-{i:02d}/data.h5"))
        metas.append(md)
        for b in (2, 3):  # two batches per case
            store.add_cells({Variable.U: torch.randn(b, n_cells, 3, generator=g), Variable.P: torch.randn(b, n_cells, 1, generator=g)}, md)
    return store, metas, V


def test_sample_store_hdf5_layout_over_the_h5py_interface():
    """SampleStore.save_h5 writes the reference's samples file (metrics.py:60-88): <case>/data/<variable> resizable
    datasets (n_samples, n_cells, dims), one sample per chunk, attribute n_samples, appending to cases the file already
    holds; from_h5 reads it back.  Through tests/h5fake.py (h5py's File / Group / Dataset interface; the image has no
    h5py) -- the same code runs over real files in test_sample_store_and_reader_over_real_hdf5_files."""
    import h5fake
    from turbdiff_amd.models.metrics import SampleStore

    store, metas, V = _store_with_samples()
    path = "/fake/samples/val.h5"
    h5fake.TREES.pop(path, None)
    store.save(path, opener=h5fake.File)
    root = h5fake.TREES[path]
    assert sorted(root.keys()) == sorted(md.case_name for md in metas)
    for md in metas:
        dg = root[md.case_name]["data"]
        assert dg.attrs["n_samples"] == 5 and sorted(dg.keys()) == ["p", "u"]
        assert dg["u"].shape == (5, md.cell_idx.numel(), 3) and dg["p"].shape == (5, md.cell_idx.numel(), 1)
        assert dg["u"].chunks == (1, md.cell_idx.numel(), 3) and dg["u"].maxshape == (None, md.cell_idx.numel(), 3)
        assert dg["u"].data.dtype == np.float32
    # a second save appends (the reference's add_samples grows the datasets)
    store.save_h5(path, opener=h5fake.File)
    assert root[metas[0].case_name]["data"].attrs["n_samples"] == 10 and root[metas[0].case_name]["data"]["u"].shape[0] == 10
    back = SampleStore.from_h5(path, V, opener=h5fake.File)
    for md in metas:
        a, b = back.load_samples(md).samples, store.load_samples(md).samples
        for v in V:
            assert torch.equal(a[v][:5], b[v]) and torch.equal(a[v][5:], b[v])
    assert back.load_samples(metas[1], range=3).samples[V[0]].shape == (1, 50, 3)


def test_sample_store_and_reader_over_real_hdf5_files(tmp_path):
    """Wherever h5py exists (not in the build image): (i) the samples file round-trips through real HDF5 and h5py itself
    sees the reference's layout; (ii) OpenFOAMDataRepository with its DEFAULT opener reads a real data.h5 written from
    the same seeded case tree the in-memory stand-in serves, and returns the same metadata and rows."""
    h5py = pytest.importorskip("h5py")
    import h5fake
    from turbdiff_amd.data.ofles import OpenFOAMDataRepository, Variable
    from turbdiff_amd.models.metrics import SampleStore

    store, metas, V = _store_with_samples(seed=4)
    path = tmp_path / "samples" / "test.h5"
    store.save(path)
    store.save_h5(path)
    with h5py.File(path, "r") as f:
        ds = f[metas[0].case_name]["data"]["u"]
        assert ds.shape == (10, 37, 3) and ds.chunks == (1, 37, 3) and ds.maxshape == (None, 37, 3)
        assert f[metas[0].case_name]["data"].attrs["n_samples"] == 10
    back = SampleStore.from_h5(path, V)
    assert torch.equal(back.load_samples(metas[1]).samples[V[1]][:5], store.load_samples(metas[1]).samples[V[1]])
    # (ii) the case-file reader over a real file vs over the stand-in
    tree = h5fake.make_case(3, counts=(9, 7, 6), n_times=12)
    real = tmp_path / "train" / "case-00" / "data.h5"
    real.parent.mkdir(parents=True)
    h5fake.dump_to_h5py(tree, real)
    h5fake.TREES[str(real)] = tree
    vars_ = (Variable.U, Variable.P, Variable.NUT)
    a = OpenFOAMDataRepository([real], vars_)                      # h5py.File
    b = OpenFOAMDataRepository([real], vars_, opener=h5fake.File)   # the stand-in
    assert np.array_equal(a.times[0], b.times[0])
    idx = [7, 2, 2, 11, 0]  # unsorted with a duplicate: the reader sorts / uniquifies for HDF5 and spreads back
    da, db = a.read(0, idx), b.read(0, idx)
    for v in vars_:
        assert torch.equal(da.samples[v], db.samples[v])
    assert torch.equal(da.metadata.cell_idx, db.metadata.cell_idx) and da.metadata.nu == db.metadata.nu
    assert sorted(da.metadata.boundaries) == sorted(db.metadata.boundaries)


@pytest.mark.gpu
def test_graphed_training_step_equals_the_eager_step_and_serves_other_geometries():
    """forward + backward of DiffusionTrainer.training_step as ONE captured hipGraph
    (training.GraphedTrainingStep).  (a) With injected timesteps / noise the replayed step gives the eager step's loss
    and gradients -- every U-Net parameter AND the learned cell-type table, whose gradient leaves the graph through the
    conditioning tensor and is chained eagerly -- for two geometries of one grid size with different in-domain cell
    counts through the SAME graph (n_cells is read on the device: tdx_masked_loss_dyn); (b) the default mode
    (enable_graph_step: t and noise drawn inside the graph) trains: losses differ from replay to replay, parameters move,
    no re-capture; (c) weights packed inside the graph: after optimiser steps the replay uses the NEW weights."""
    from turbdiff_amd.data.ofles import BoundaryCondition as BC
    from turbdiff_amd.training import DiffusionTrainer, GraphedTrainingStep

    gen = torch.Generator().manual_seed(1)
    counts = (12, 10, 9)
    idx = torch.arange(12 * 10 * 9).reshape(counts)

    def case(hole):
        inside = torch.zeros(counts, dtype=torch.bool)
        inside[1:-1, 1:-1, 1:-1] = True
        if hole:
            inside[3:6, 2:5, 2:6] = False
        cell_idx = inside.flatten().nonzero().flatten()
        meta = OpenFOAMMetadata(np.array(counts), cell_idx, {"walls": {"idx": idx[:, 0].flatten()}, "inlets": {"idx": idx[0].flatten()}},
                                {Variable.U: {"inlets": BC(BC.Type.FIXED_VALUE, torch.tensor([1.0, 0.0, 0.0]))}})
        fields = {Variable.U: torch.randn(4, len(cell_idx), 3, generator=gen), Variable.P: torch.randn(4, len(cell_idx), 1, generator=gen)}
        return meta, np.arange(4) * 0.1, fields

    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3) * 1.5}, "p": {"mean": torch.tensor(0.1), "std": torch.tensor(0.8)}})
    ds = OpenFOAMDataset(InMemoryRepository([case(False), case(True)]), stats, discard_first_seconds=-1.0)
    dev = torch.device("cuda:0")
    to_dev = lambda b: next(iter(__import__("turbdiff_amd.data.staging", fromlist=["DeviceStager"]).DeviceStager([b], dev)))
    batches = [to_dev(ds[[0, 1]]), to_dev(ds[[4, 5]])]  # one batch per geometry
    assert batches[0].data.metadata.cell_idx.numel() != batches[1].data.metadata.cell_idx.numel()
    torch.manual_seed(0)
    tr = DiffusionTrainer(**{**DiffusionTrainer.SHIPPED_CONFIG, "dim": 8, "timesteps": 10}, u_net_levels=2, normalization_mode="mean-std",
                          max_train_steps=10, compute_mode="f32").to(dev)
    table = tr.cell_type_embedding.embedding.weight
    # (a)
    gs = GraphedTrainingStep(tr, inject=True)
    for i, b in enumerate(batches):
        t = torch.tensor([3, 7], device=dev)
        noise = torch.randn(2, 4, *counts, generator=torch.Generator().manual_seed(10 + i)).to(dev)
        tr.zero_grad(set_to_none=True)
        x, C = tr._model_input(b)
        from types import SimpleNamespace
        loss, _ = tr.model.p_losses(x, t, C, SimpleNamespace(cell_idx=b.data.metadata.cell_idx), None, noise=noise)
        loss.backward()
        want = {n: p.grad.clone() for n, p in tr.named_parameters() if p.grad is not None}
        assert "cell_type_embedding.embedding.weight" in want or "conditioning.cell_type_embedding.embedding.weight" in want
        want_loss = loss.item()
        del loss, x, C  # no eager autograd graph may be alive when the step is captured (its AccumulateGrad nodes
        tr.zero_grad(set_to_none=True)  # belong to the stream the eager backward ran on)
        gs.set_draws(t, noise)
        got = gs(b)
        assert len(gs.slots) == 1, "both geometries must replay one graph"
        assert abs(got.item() - want_loss) < 1e-5 * abs(want_loss)
        for n, p in tr.named_parameters():
            if n in want:
                assert p.grad is not None, n
                d = (p.grad - want[n]).norm().item()
                assert d <= 1e-4 * want[n].norm().item() + 1e-9, (i, n, d)
    assert table.grad is not None and table.grad.abs().sum() > 0
    # (b), (c)
    tr.zero_grad(set_to_none=True)
    tr.enable_graph_step()
    before = {n: p.detach().clone() for n, p in tr.named_parameters()}
    torch.manual_seed(5)
    losses = [tr.fit_step(batches[i % 2]).item() for i in range(6)]
    assert all(np.isfinite(losses)) and len(set(round(l, 6) for l in losses)) > 1
    assert len(tr._graph_step.slots) == 1
    moved = [n for n, p in tr.named_parameters() if not torch.equal(p.detach(), before[n])]
    # (a conv bias in front of a GroupNorm has a mathematically zero gradient: rounding noise may or may not move it)
    assert all(".conv.bias" in n for n in set(before) - set(moved)) and len(moved) > 0.9 * len(before), set(before) - set(moved)
    # the graph packs the weights of the moment: an eager step on the same inputs and draws agrees with a replay after
    # six optimiser steps
    gs2 = GraphedTrainingStep(tr, inject=True)
    t = torch.tensor([1, 9], device=dev)
    noise = torch.randn(2, 4, *counts, generator=torch.Generator().manual_seed(77)).to(dev)
    gs2.set_draws(t, noise)
    gs2(batches[0])               # captures (and replays once) under the current weights
    tr._opt.step()                # weights change in place
    got = gs2(batches[0]).item()
    x, C = tr._model_input(batches[0])
    with torch.no_grad():
        want, _ = tr.model.p_losses(x, t, C, SimpleNamespace(cell_idx=batches[0].data.metadata.cell_idx), None, noise=noise)
    assert abs(got - want.item()) < 1e-5 * abs(want.item()), (got, want.item())
