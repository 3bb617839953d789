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
