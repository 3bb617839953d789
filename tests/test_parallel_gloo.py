"""World-size-2 tests of the data-parallel logic on CPU (gloo): bucketed, hook-driven gradient
averaging equals the single-process gradient of the concatenated batch; trajectory sharding
covers every trajectory exactly once."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.SiLU(), torch.nn.Linear(32, 32), torch.nn.SiLU(),
                               torch.nn.Linear(32, 3))


def _worker(rank, world, port, outdir, variant="dynamic"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "generative-turbulence_amd"))
    from turbdiff_amd.parallel import BucketedDataParallel, init_from_env

    init_from_env("gloo")
    model = _make_model()
    if rank != 0:  # de-synchronise: the wrapper must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    if variant != "dynamic":
        # a module that knows its backward order (as models.ddpm.DenoisingModel does): buckets exist -- and
        # all-reduces overlap -- from the first step on
        model.grad_ready_order = lambda: reversed(list(model.parameters()))
    ddp = BucketedDataParallel(model, bucket_mb=0.003,  # ~3 KB buckets -> several buckets
                               compress="bf16" if variant == "bf16" else None)
    if variant != "dynamic":
        assert ddp.bucket_layout() is not None, "static order must give buckets before the first backward"
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, 8, 6, generator=g)  # 3 steps, global batch 8
    out = []
    for step in range(3):
        x = data[step, rank * 4 : (rank + 1) * 4]
        model.zero_grad(set_to_none=True)
        model(x).pow(2).mean().backward()
        ddp.finish()
        out.append([p.grad.clone() for p in model.parameters()])
        flats = {f.data_ptr(): f for f in ddp._flat}
        assert all(any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + f.numel() * 4 for f in flats.values())
                   for p in model.parameters()), "gradients must be views into the persistent buckets"
    if variant == "static":  # a second backward before finish() is a contract violation, reported loudly
        model.zero_grad(set_to_none=True)
        model(data[0, :4]).pow(2).mean().backward()
        try:
            model(data[0, :4]).pow(2).mean().backward()
            raise AssertionError("second backward before finish() must raise")
        except RuntimeError as e:
            assert "one backward per finish" in str(e)
        ddp._seen.clear(); ddp._launched.clear(); ddp._work.clear(); ddp._pending = [len(b) for b in ddp._buckets]
    torch.save((out, ddp.bucket_layout(), [p.detach().clone() for p in model.parameters()]), f"{outdir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("variant", ["dynamic", "static", "bf16"])
def test_bucketed_ddp_matches_single_process(tmp_path, variant):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), variant)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=100)
        assert p.exitcode == 0
    results = {r: torch.load(tmp_path / f"rank{r}.pt") for r in range(2)}
    # reference: one process, global batch
    model = _make_model()
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, 8, 6, generator=g)
    for rank in (0, 1):
        for a, b in zip(results[rank][2], model.parameters()):
            assert torch.equal(a, b), "parameters were not broadcast from rank 0"
    for step in range(3):
        model.zero_grad(set_to_none=True)
        # mean over the global batch == mean of the two per-rank means (equal shard sizes)
        model(data[step]).pow(2).mean().backward()
        for rank in (0, 1):
            for gr, p in zip(results[rank][0][step], model.parameters()):
                if variant == "bf16":  # gradients travelled as bfloat16
                    assert torch.allclose(gr, p.grad, rtol=2e-2, atol=1e-3)
                else:
                    assert torch.allclose(gr, p.grad, rtol=1e-5, atol=1e-7)
    layout = results[0][1]
    assert layout is not None and len(layout) >= 2, layout
    assert sum(n for n, _ in layout) == len(list(model.parameters()))


def test_shard_trajectories_partition():
    from turbdiff_amd.parallel import shard_trajectories

    for n, world in [(64, 8), (10, 4), (3, 8), (64, 1)]:
        seen = []
        for r in range(world):
            seen += list(shard_trajectories(n, r, world))
        assert seen == list(range(n))
    assert len(shard_trajectories(64, 3, 8)) == 8


# --------------------------------------------------------------------------- sharded data feed, world size 2


def _unequal_cases():
    """Two geometries with unequal sample counts (5 and 3 usable time steps): at batch size 2 that is 3 + 2 = 5 batch
    lists -- an odd number, so the two ranks' shards only have equal length because the sampler wraps around."""
    import numpy as np

    from turbdiff_amd.data.ofles import OpenFOAMMetadata, Variable

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


def _feed_worker(rank, world, port, outdir, device):
    """One rank of a data-parallel epoch: its shard of the batch lists (OpenFOAMSampler(rank, world)), optionally
    staged to the device (DeviceStager), one all-reduce per batch as the gradient exchange would do -- a rank with
    fewer batches than its peer would leave the other hanging in the collective (the test's timeout)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "generative-turbulence_amd"))
    from turbdiff_amd.data.ofles import InMemoryRepository, OpenFOAMDataset, OpenFOAMSampler, OpenFOAMStats, Variable
    from turbdiff_amd.parallel import init_from_env

    init_from_env("gloo")
    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3)}, "p": {"mean": torch.tensor(0.0), "std": torch.tensor(1.0)}})
    ds = OpenFOAMDataset(InMemoryRepository(_unequal_cases()), stats, discard_first_seconds=-1.0)
    log = []
    for epoch in range(2):
        s = OpenFOAMSampler(ds, batch_size=2, shuffle=True, rank=rank, world_size=world, seed=5)
        s.set_epoch(epoch)
        lists = list(s)
        feed = (ds[b] for b in lists)
        if device != "cpu":
            from turbdiff_amd.data.staging import DeviceStager

            feed = DeviceStager(feed, device)
        sums = []
        for b in feed:
            u = b.data.samples[Variable.U]
            assert u.device.type == torch.device(device).type
            v = u.float().sum().reshape(1).cpu()
            dist.all_reduce(v)  # lock-step: every rank must arrive here the same number of times
            sums.append((v.item(), tuple(int(c) for c in b.data.metadata.cell_counts) if hasattr(b.data.metadata, "cell_counts") else None))
        log.append((lists, sums))
    torch.save(log, f"{outdir}/feed{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _check_feed(tmp_path):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "generative-turbulence_amd"))
    from turbdiff_amd.data.ofles import InMemoryRepository, OpenFOAMDataset, OpenFOAMSampler, OpenFOAMStats, Variable

    logs = [torch.load(tmp_path / f"feed{r}.pt") for r in range(2)]
    stats = OpenFOAMStats({"u": {"mean": torch.zeros(3), "std": torch.ones(3)}, "p": {"mean": torch.tensor(0.0), "std": torch.tensor(1.0)}})
    ds = OpenFOAMDataset(InMemoryRepository(_unequal_cases()), stats, discard_first_seconds=-1.0)
    for epoch in range(2):
        full = OpenFOAMSampler(ds, batch_size=2, shuffle=True, seed=5)
        full.set_epoch(epoch)
        ref = list(full)
        assert len(ref) == 5  # 3 + 2 batch lists: odd
        a, b = logs[0][epoch][0], logs[1][epoch][0]
        assert len(a) == len(b) == 3, "both ranks must see the same number of batches (the odd one wraps around)"
        merged = [x for pair in zip(a, b) for x in pair]
        assert merged[:5] == ref and merged[5] in ref
        # the all-reduced sums: both ranks hold the same value per step = sum over the two ranks' batches
        for i, ((va, _), (vb, _)) in enumerate(zip(logs[0][epoch][1], logs[1][epoch][1])):
            want = sum(ds[x].data.samples[Variable.U].float().sum().item() for x in (a[i], b[i]))
            assert va == vb and abs(va - want) < 1e-3 * max(1.0, abs(want))
    assert logs[0][0][0] != logs[0][1][0]  # epochs reshuffle


@pytest.mark.timeout(120)
def test_sharded_sampler_feeds_two_ranks_in_lock_step(tmp_path):
    """OpenFOAMSampler under world size 2 with unequal case counts (an odd number of batch lists):
    both ranks iterate the same number of single-geometry batches, together they cover the single-process order, and a
    collective per batch completes."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_feed_worker, args=(r, 2, port, str(tmp_path), "cpu")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=100)
        assert p.exitcode == 0
    _check_feed(tmp_path)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_sampler_and_device_stager_two_ranks_one_gpu(tmp_path):
    """The same epoch with each rank's shard staged through its own DeviceStager (pinned buffers + copy stream) on
    cuda:0: two processes, one device."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_feed_worker, args=(r, 2, port, str(tmp_path), "cuda:0")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=280)
        assert p.exitcode == 0
    _check_feed(tmp_path)
