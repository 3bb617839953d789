"""Two ranks sharing the one GPU of the test box (gloo backend on device tensors): the
hook-driven bucketed all-reduce works with the HIP autograd Functions, and the averaged
gradients equal the single-process gradients of the global batch."""

import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(dev):
    sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
    from turbdiff_amd.models.conditioning import Conditioning
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    g = np.load(ROOT / "tests" / "golden" / "model_cfg1.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=8,
                         u_net_levels=2, norm_type="group")
    net.load_state_dict(sd)
    diff = GaussianDiffusion(net, timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True).to(dev)
    x = torch.from_numpy(g["x"]).to(dev)            # global batch of 2
    C = {Conditioning.Type.CELL_TYPE: torch.from_numpy(g["c_local"]).to(dev)}
    md = SimpleNamespace(cell_idx=torch.from_numpy(g["cell_idx"]).to(dev))
    t = torch.from_numpy(g["t"]).to(dev)
    noise = torch.from_numpy(g["loss_nb1/noise"]).to(dev)
    return diff, x, C, md, t, noise


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dev = torch.device("cuda:0")
    diff, x, C, md, t, noise = _build(dev)
    from turbdiff_amd.parallel import BucketedDataParallel, init_from_env

    init_from_env("gloo")
    ddp = BucketedDataParallel(diff, bucket_mb=0.25)
    grads = []
    for step in range(2):  # step 0 discovers the ready order, step 1 overlaps
        diff.zero_grad(set_to_none=True)
        loss, _ = diff.p_losses(x[rank : rank + 1], t[rank : rank + 1], C, md, None, noise=noise[rank : rank + 1])
        loss.backward()
        ddp.finish()
        grads.append({n: p.grad.detach().cpu().clone() for n, p in diff.model.named_parameters()})
    # the fused optimiser on the all-reduced gradients (views into the flat buckets)
    from turbdiff_amd.optim import ClipRAdam

    opt = ClipRAdam(diff.parameters(), lr=1e-3, max_norm=0.1)
    opt.step()
    params = {n: p.detach().cpu().clone() for n, p in diff.model.named_parameters()}
    torch.save((grads, ddp.bucket_layout(), params, float(opt.last_grad_norm)), f"{outdir}/rank{rank}.pt")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu_gradient_average(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=280)
        assert p.exitcode == 0
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    diff, x, C, md, t, noise = _build(torch.device("cuda:0"))
    loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)  # mean over the global batch of 2
    loss.backward()
    layout = res[0][1]
    assert layout is not None and len(layout) >= 2
    for name, p in diff.model.named_parameters():
        ref = p.grad.cpu()
        for r in range(2):
            for step in range(2):
                got = res[r][0][step][name]
                if ref.norm() < 1e-6:
                    assert got.norm() < 1e-5, name
                else:
                    assert ((got - ref).norm() / ref.norm()).item() < 2e-3, (name, r, step)
        assert torch.equal(res[0][0][1][name], res[1][0][1][name]), f"{name}: ranks disagree"
    # one clip + RAdam step on the global-batch gradients with the stock torch calls
    ref_norm = torch.nn.utils.clip_grad_norm_(diff.parameters(), 0.1)
    torch.optim.RAdam(diff.parameters(), lr=1e-3).step()
    for r in range(2):
        assert abs(res[r][3] - ref_norm.item()) < 2e-3 * ref_norm.item()
        for name, p in diff.model.named_parameters():
            assert torch.allclose(res[r][2][name], p.detach().cpu(), rtol=1e-4, atol=2e-6), (name, r)


def _rccl_worker(port, outdir):
    """World size 1 over backend "nccl" (= RCCL): communicator creation, the hook-driven bucketed all-reduce of
    device buffers on RCCL's stream, finish() -- everything the 2/4/8-GPU runs execute except a second peer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    diff, x, C, md, t, noise = _build(dev)
    from turbdiff_amd.parallel import BucketedDataParallel, init_from_env

    rank, world, local = init_from_env("nccl", force=True)
    assert (rank, world) == (0, 1) and torch.distributed.get_backend() == "nccl"
    ddp = BucketedDataParallel(diff, bucket_mb=0.25, force=True)
    assert ddp.active and ddp.bucket_layout() is not None  # static order from DenoisingModel.grad_ready_order()
    res = {}
    for tag, compress in (("f32", None), ("bf16", "bf16")):
        ddp.compress = compress
        if compress:
            ddp._wire = [torch.zeros_like(f, dtype=torch.bfloat16) for f in ddp._flat]
        diff.zero_grad(set_to_none=True)
        loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
        loss.backward()
        ddp.finish()
        res[tag] = {n: p.grad.detach().cpu().clone() for n, p in diff.model.named_parameters()}
    ddp.allreduce_only()
    torch.save((res, ddp.bucket_layout(), ddp.stats), f"{outdir}/rccl.pt")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_world_size_one_runs_the_communicator_path(tmp_path):
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(timeout=280)
    assert p.exitcode == 0
    res, layout, stats = torch.load(tmp_path / "rccl.pt")
    assert len(layout) >= 2 and stats["steps"] == 2
    diff, x, C, md, t, noise = _build(torch.device("cuda:0"))
    loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
    loss.backward()
    for name, p in diff.model.named_parameters():
        ref = p.grad.cpu()
        if ref.norm() < 1e-6:
            continue
        assert ((res["f32"][name] - ref).norm() / ref.norm()).item() < 2e-3, name     # atomics: not bit-identical
        assert ((res["bf16"][name] - ref).norm() / ref.norm()).item() < 1e-2, name    # travelled as bfloat16
