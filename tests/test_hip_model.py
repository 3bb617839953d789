"""End-to-end parity of the HIP hot path against the golden vectors generated from the
reference (tests/golden) and, at the full BASELINE size, against the CPU oracle.

Tolerances (rel-L2): fp32 mode 1e-4 on outputs (north_star gate), 1e-3 on gradients;
bf16 mode (bf16 activation storage + bf16 MFMA operands, fp32 accumulation) 3e-2.
"""

from types import SimpleNamespace

import pytest
import torch

from conftest import assert_grad_close, rel_l2
from oracle import turbdiff_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def build_cfg1(golden, noise_bcs=True, dtype=torch.float32, sd=None):
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=8,
                         u_net_levels=2, norm_type="group")
    net.load_state_dict(sd if sd is not None else golden("model_cfg1").sub("sd/"), strict=True)
    net.set_compute_dtype(dtype)
    return GaussianDiffusion(net, timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=noise_bcs).to(dev())


def cond(c_local):
    from turbdiff_amd.models.conditioning import Conditioning

    return {Conditioning.Type.CELL_TYPE: c_local.to(dev())}


def test_schedule_buffers_match_golden_on_device(golden):
    g = golden("schedules")
    diff = build_cfg1(golden)
    for k in ["betas", "posterior_log_var", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef2"]:
        assert torch.equal(getattr(diff, k).cpu(), g[f"log-snr-linear/10/{k}"])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_denoiser_forward_golden(golden, dtype, tol):
    g = golden("model_cfg1")
    diff = build_cfg1(golden, dtype=dtype)
    with torch.no_grad():
        y = diff.model(g["x"].to(dev()), g["t"].to(dev()), cond(g["c_local"]))
    assert y.shape == g["eps_hat"].shape and y.dtype == torch.float32
    assert rel_l2(y.cpu(), g["eps_hat"]) < tol
    g2 = golden("model_cfg1_48")
    with torch.no_grad():
        y = diff.model(g2["x"].float().to(dev()), g2["t"].to(dev()), cond(g2["c_local"].float()))
    assert rel_l2(y.cpu(), g2["eps_hat"]) < tol


@pytest.mark.parametrize("nb", [1, 0])
def test_p_losses_and_grads_golden(golden, nb):
    g = golden("model_cfg1")
    diff = build_cfg1(golden, noise_bcs=bool(nb))
    loss, t = diff.p_losses(g["x"].to(dev()), g["t"].to(dev()), cond(g["c_local"]),
                            SimpleNamespace(cell_idx=g["cell_idx"].to(dev())), None, noise=g[f"loss_nb{nb}/noise"].to(dev()))
    loss.backward()
    ref = g[f"loss_nb{nb}/loss"].item()
    assert abs(loss.item() - ref) < 1e-4 * abs(ref)
    n = 0
    for name, p in diff.model.named_parameters():
        assert p.grad is not None, name
        assert_grad_close(name, p.grad.cpu(), g[f"loss_nb{nb}/grad/{name}"], 1e-3)
        n += 1
    assert n == len(g.keys(f"loss_nb{nb}/grad/"))


def test_p_losses_bf16_close(golden):
    g = golden("model_cfg1")
    diff = build_cfg1(golden, dtype=torch.bfloat16)
    loss, _ = diff.p_losses(g["x"].to(dev()), g["t"].to(dev()), cond(g["c_local"]),
                            SimpleNamespace(cell_idx=g["cell_idx"].to(dev())), None, noise=g["loss_nb1/noise"].to(dev()))
    loss.backward()
    ref = g["loss_nb1/loss"].item()
    assert abs(loss.item() - ref) < 2e-2 * abs(ref)
    w = "u_net.downsampling_blocks.1.block2.conv.weight"
    p = dict(diff.model.named_parameters())[w]
    assert rel_l2(p.grad.cpu(), g[f"loss_nb1/grad/{w}"]) < 0.1


@pytest.mark.parametrize("tag,nb,start", [("nb1", True, None), ("nb0", False, None), ("nb1_from5", True, 5)])
def test_p_sample_loop_golden(golden, tag, nb, start):
    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=nb)
    noises = [g[f"{tag}/noise/{i}"].to(dev()) for i in range(int(g[f"{tag}/n_noise"]))]
    it = iter(noises)
    out = diff.p_sample_loop(g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev()), start_from=start,
                             noise_fn=lambda like: next(it))
    assert next(it, None) is None, "sampler drew fewer noise tensors than the reference"
    assert rel_l2(out.cpu(), g[f"{tag}/out"]) < 1e-4
    # boundary / outside cells are clamped to the data values at the end (ddpm.py:814)
    inside = torch.zeros(out[0, 0].numel(), dtype=torch.bool)
    inside[g["cell_idx"]] = True
    assert torch.equal(out.cpu().flatten(-3)[..., ~inside], g["x_bcs"].flatten(-3)[..., ~inside])


def test_p_sample_mean_logvar_golden(golden):
    g = golden("sample_cfg1")
    diff = build_cfg1(golden)
    mean, log_var = diff.p_sample(g["x_bcs"].to(dev()), 6, cond(g["c_local"]), g["cell_idx"].to(dev()))
    assert rel_l2(mean.cpu(), g["p_sample_t6/mean"]) < 1e-4
    assert torch.equal(log_var.cpu(), g["p_sample_t6/log_var"])


@pytest.mark.parametrize("fused", [False, True])
def test_three_training_steps_golden(golden, fused):
    """clip 0.1 -> RAdam(1e-4) -> exp LambdaLR, as DiffusionTraining.configure_optimizers
    (diffusion.py:210-235) with trainer.gradient_clip_val = 0.1 (train.yaml:30-31); with the stock
    torch calls and with the fused ClipRAdam."""
    import math

    from turbdiff_amd.optim import ClipRAdam

    g = golden("train_cfg1")
    diff = build_cfg1(golden)
    opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1) if fused else torch.optim.RAdam(diff.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: math.exp(math.log(1e-6 / 1e-4) / 20 * min(s, 20)))
    x, C, md = g["x"].to(dev()), cond(g["c_local"]), SimpleNamespace(cell_idx=g["cell_idx"].to(dev()))
    for step in range(3):
        loss, _ = diff.p_losses(x, g[f"step{step}/t"].to(dev()), C, md, None, noise=g[f"step{step}/noise"].to(dev()))
        opt.zero_grad()
        loss.backward()
        if fused:
            opt.step()
            gn = opt.last_grad_norm
        else:
            gn = torch.nn.utils.clip_grad_norm_(diff.parameters(), 0.1)
            opt.step()
        sched.step()
        assert abs(loss.item() - g[f"step{step}/loss"].item()) < 2e-4 * abs(g[f"step{step}/loss"].item())
        assert abs(gn.item() - g[f"step{step}/grad_norm"].item()) < 2e-3 * g[f"step{step}/grad_norm"].item()
        assert abs(sched.get_last_lr()[0] - g[f"step{step}/lr_after"].item()) < 1e-12
    for k, v in diff.model.state_dict().items():
        assert rel_l2(v.cpu(), g[f"final_sd/{k}"]) < 1e-5, k


def test_full_size_forward_matches_oracle_fp32(monkeypatch):
    """BASELINE config 2 geometry (192x64x48, dim 32, 4 levels, GN(8)), B = 1, fp32 mode,
    default-initialised weights (seed 0) vs the CPU oracle: the 1e-4 rel-L2 gate."""
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(0)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                         u_net_levels=4, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g1, g2 = torch.Generator().manual_seed(1234), torch.Generator().manual_seed(1235)
    x = torch.randn(1, 4, 192, 64, 48, generator=g1)
    c_local = torch.randn(4, 192, 64, 48, generator=g2)
    t = torch.tensor([123])
    with torch.no_grad():
        ref = O.denoiser(sd, x, t, c_local, timesteps=500)
        net.to(dev())
        y = net(x.to(dev()), t.to(dev()), cond(c_local))
    assert rel_l2(y.cpu(), ref) < 1e-4
    # fp32 tensors with split-precision (bf16 hi + lo, 3 MFMAs per product) convs: same 1e-4 gate
    monkeypatch.setenv("TDX_CONV_IMPL", "split")
    with torch.no_grad():
        ys = net(x.to(dev()), t.to(dev()), cond(c_local))
    monkeypatch.delenv("TDX_CONV_IMPL")
    assert rel_l2(ys.cpu(), ref) < 1e-4 and not torch.equal(ys, y)
    net.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        yb = net(x.to(dev()), t.to(dev()), cond(c_local))
    assert rel_l2(yb.cpu(), ref) < 3e-2
    # fp16 storage + fp16 MFMA operands (11 significand bits, the precision of the reference's TF32 GPU convs): the same
    # kernels as bf16, an order of magnitude closer to the oracle (<= 2e-3; bf16 measures 8.9e-3)
    net.set_compute_dtype(torch.float16)
    with torch.no_grad():
        yh = net(x.to(dev()), t.to(dev()), cond(c_local))
    assert rel_l2(yh.cpu(), ref) < 2e-3, rel_l2(yh.cpu(), ref)
    assert rel_l2(yh.cpu(), ref) < 0.5 * rel_l2(yb.cpu(), ref)


def _full_size_problem(seed=0):
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(seed)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                         u_net_levels=4, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return net, sd


@pytest.mark.timeout(900)
def test_full_size_gradients_match_oracle(monkeypatch):
    """BASELINE config 2 at full size (192x64x48, dim 32, 4 levels, B = 1): loss and parameter gradients of
    p_losses against the CPU oracle's backward, in the fp32 mode, with split-precision convs and in the bf16 headline
    mode.  This is where the level-0 split-K weight gradient, the original-grid data gradient + halo-shell kernel and the
    ring / brick tiling of the finest level run at the sizes the benchmark uses.  Tolerance 1e-3 rel-L2 per tensor in
    the fp32 modes (north_star: fp32), 0.1 in bf16."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    net, sd = _full_size_problem()
    X, Y, Z = 192, 64, 48
    x = torch.randn(1, 4, X, Y, Z, generator=torch.Generator().manual_seed(1234))
    c_local = torch.randn(4, X, Y, Z, generator=torch.Generator().manual_seed(1235))
    noise = torch.randn(1, 4, X, Y, Z, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([250])
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[13:25, 24:40, 0:32] = False
    cell_idx = m.flatten().nonzero().flatten()
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 500)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=500, noise_bcs=True)
    ref_loss.backward()
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    md = SimpleNamespace(cell_idx=cell_idx.to(dev()))
    watched = WATCHED_FULL_SIZE
    _check_modes_against_oracle(monkeypatch, diff, (x, t, c_local, noise), md, ref_loss, leaves, watched)


# every level's convs, the bottleneck attention, both encoders (through the composed first conv), FiLM and norms
WATCHED_FULL_SIZE = [
    "u_net.downsampling_blocks.0.block1.conv.weight", "u_net.downsampling_blocks.0.block2.conv.weight",
    "u_net.downsampling_blocks.1.block1.conv.weight", "u_net.downsampling_blocks.3.block2.conv.weight",
    "u_net.center_block.0.block1.conv.weight", "u_net.center_block.1.fn.fn.to_qkv.weight",
    "u_net.upsampling_blocks.0.block1.conv.weight", "u_net.upsampling_blocks.3.block1.conv.weight",
    "u_net.upsampling_blocks.3.conv.weight", "decode.0.block2.conv.weight", "decode.1.weight", "encode_x.weight",
    "encode_c_local.weight", "encode_x.bias", "u_net.downsampling_blocks.0.project_onto_scale_shift.weight",
    "u_net.downsampling_blocks.0.block1.norm.weight", "u_net.upsampling_blocks.3.block2.norm.bias", "process_c.0.weight"]


def _check_modes_against_oracle(monkeypatch, diff, inputs, md, ref_loss, leaves, watched):
    """loss + watched parameter gradients of p_losses in the three arithmetic modes against the oracle's: IEEE fp32 MFMA
    convs and split-precision convs on fp32 tensors at 1e-3 (north_star: fp32 parity), bf16 storage (the benchmark's
    headline mode: ring / brick conv kernels, level-0 split-K weight gradient with atomics, bf16 halo-shell atomics) at
    0.1 per tensor and 3e-2 on the loss."""
    x, t, c_local, noise = inputs
    from turbdiff_amd.training import DiffusionTrainer

    # fp16 (round 6): fp16 tensors / MFMA operands under a power-of-two loss scale (the backward pass of S * loss; the
    # fp32 parameter gradients come out S times too large and are divided here, as ClipRAdam does): 2e-2 per tensor, 5x
    # inside the bf16 gate, and 2e-3 on the loss
    n_loss = x.shape[0] * x.shape[1] * int(md.cell_idx.numel())
    for mode, impl, dtype, tol, ltol in (("f32", "auto", torch.float32, 1e-3, 1e-4), ("f32s", "split", torch.float32, 1e-3, 1e-4),
                                         ("bf16", "auto", torch.bfloat16, 0.1, 3e-2), ("fp16", "auto", torch.float16, 2e-2, 2e-3)):
        monkeypatch.setenv("TDX_CONV_IMPL", impl)
        diff.model.set_compute_dtype(dtype)
        diff.zero_grad(set_to_none=True)
        S = DiffusionTrainer.initial_loss_scale(n_loss) if mode == "fp16" else 1.0
        loss, _ = diff.p_losses(x.to(dev()), t.to(dev()), cond(c_local), md, None, noise=noise.to(dev()))
        (loss * S).backward()
        assert abs(loss.item() - ref_loss.item()) < ltol * abs(ref_loss.item()), (mode, loss.item(), ref_loss.item())
        params = dict(diff.model.named_parameters())
        for name in watched:
            g = params[name].grad.float().cpu() / S
            assert torch.isfinite(g).all(), (mode, name)
            assert_grad_close(f"{mode}:{name}", g, leaves[name].grad, tol)
    monkeypatch.delenv("TDX_CONV_IMPL")
    diff.model.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("mode,impl,dtype", [("bf16", "auto", torch.bfloat16), ("fp16", "auto", torch.float16),
                                             ("f32s", "split", torch.float32), ("f32", "auto", torch.float32)])
@pytest.mark.parametrize("grid,B", [((96, 32, 24), 2), ((192, 64, 48), 3), ((50, 26, 22), 2)])
def test_deterministic_switch_gives_bit_identical_gradients(grid, B, mode, impl, dtype, monkeypatch):
    """TDX_DETERMINISTIC=1: three backward passes of the same training step leave the SAME BITS in all 139 parameter
    gradients (by default 136 of them differ run to run: halo-shell atomics, and fp32 atomic merges of the bias / 1x1 /
    encoder / decoder / first-conv gradients) -- at a small grid, at the benchmark's grid (ring / producer-consumer kernels,
    many K splits) and at a ragged one -- and the gradients are the default path's to fp32 summation order."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    net, _ = _full_size_problem(seed=5)
    X, Y, Z = grid
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 4, X, Y, Z, generator=g).to(dev())
    c_local = torch.randn(4, X, Y, Z, generator=g)
    noise = torch.randn(B, 4, X, Y, Z, generator=g).to(dev())
    t = torch.tensor([3, 250, 499][:B]).to(dev())
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    md = SimpleNamespace(cell_idx=m.flatten().nonzero().flatten().to(dev()))
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    monkeypatch.setenv("TDX_CONV_IMPL", impl)
    diff.model.set_compute_dtype(dtype)
    S = 2.0**12 if mode == "fp16" else 1.0

    def grads():
        diff.zero_grad(set_to_none=True)
        loss, _ = diff.p_losses(x, t, cond(c_local), md, None, noise=noise)
        (loss * S).backward()
        torch.cuda.synchronize()
        return loss.item(), {n: p.grad.clone() for n, p in diff.model.named_parameters()}

    base_loss, base = grads()  # the default path (atomics)
    monkeypatch.setenv("TDX_DETERMINISTIC", "1")
    runs = [grads() for _ in range(3)]
    monkeypatch.delenv("TDX_DETERMINISTIC")
    # (the deterministic forward takes its GroupNorm statistics from the stored conv result, the default one from the conv
    # kernels' fp32 accumulators: the losses agree to the tensors' rounding, not bit for bit)
    assert len({r[0] for r in runs}) == 1
    assert abs(runs[0][0] - base_loss) < (2e-3 if mode in ("bf16", "fp16") else 1e-5) * abs(base_loss)
    differ = [n for r in runs[1:] for n in r[1] if not torch.equal(r[1][n], runs[0][1][n])]
    assert not differ, (mode, sorted(set(differ)))
    # the same numbers as the default path up to summation order (bf16: the shell's packed atomics round in bf16 per add,
    # the ordered route adds in fp32 -- and that difference travels down the backward pass)
    tol = {"bf16": 2e-2, "fp16": 3e-3, "f32s": 2e-5, "f32": 2e-5}[mode]
    for n, g0 in base.items():
        if g0.norm() > 0:
            assert rel_l2(runs[0][1][n].float().cpu(), g0.float().cpu()) < tol, (mode, n)
    monkeypatch.delenv("TDX_CONV_IMPL")
    diff.model.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_deterministic_switch_reproduces_a_training_run(mode, monkeypatch):
    """TDX_DETERMINISTIC=1 over whole optimiser steps: two runs of four training steps from the same weights and seeds
    (t and noise drawn on the device, clip + RAdam through ClipRAdam, weights re-packed every step; fp16 under its loss
    scale) end in bit-identical parameters and losses."""
    import bench
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    monkeypatch.setenv("TDX_DETERMINISTIC", "1")
    net, sd = _full_size_problem(seed=9)
    grid, B = (96, 32, 24), 2
    x, c_local, cell_idx = bench.synthetic_inputs(B, dev(), grid)
    md = SimpleNamespace(cell_idx=cell_idx)
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    bench.set_mode(diff, mode)
    ends = []
    for run in range(2):
        diff.model.load_state_dict(sd)
        opt = bench.new_optimizer(diff, mode, bench.LOSS_ELEMENTS(B, cell_idx))
        torch.manual_seed(123)
        losses = []
        for step in range(4):
            loss, _ = diff(x, cond(c_local), md, None)
            opt.scale_loss(loss).backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.item())
        torch.cuda.synchronize()
        ends.append((losses, {n: p.detach().clone() for n, p in diff.model.named_parameters()}))
    assert ends[0][0] == ends[1][0], (ends[0][0], ends[1][0])
    assert ends[0][0][0] != ends[0][0][3]  # the weights did move
    differ = [n for n in ends[0][1] if not torch.equal(ends[0][1][n], ends[1][1][n])]
    assert not differ, differ
    bench.set_mode(diff, "f32")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_captured_training_run_equals_the_eager_one_bit_for_bit(mode, monkeypatch):
    """With TDX_DETERMINISTIC=1 every merge of the backward pass is ordered, so 25 optimiser steps on the same (x, t, noise) from the
    same weights must end in the SAME parameter bits and losses whether forward + backward are issued eagerly (weight gradients on
    the side stream) or replayed from one captured graph (one stream, weights re-packed inside the graph, loss scale read from a
    device scalar in fp16): the captured path checked against the eager one over a whole run, not to a tolerance."""
    import bench
    from turbdiff_amd.training import GraphedTrainingStep

    monkeypatch.setenv("TDX_DETERMINISTIC", "1")
    diff = bench.build_model(dev())
    bench.set_mode(diff, mode)
    sd0 = {k: v.clone() for k, v in diff.state_dict().items()}
    B, grid = 2, (96, 32, 24)
    x, c_local, idx = bench.synthetic_inputs(B, dev(), grid)
    C, md = cond(c_local), SimpleNamespace(cell_idx=idx)
    t = torch.tensor([3, 250], device=dev())
    noise = torch.randn(x.shape, device=dev(), generator=torch.Generator(device=dev()).manual_seed(1))
    ends = {}
    for kind in ("eager", "graph"):
        diff.load_state_dict(sd0)
        diff.zero_grad(set_to_none=True)
        opt = bench.new_optimizer(diff, mode, bench.LOSS_ELEMENTS(B, idx))
        losses = []
        if kind == "graph":
            gs = GraphedTrainingStep(bench._Task(diff, opt), inject=True)
            gs.set_draws(t, noise)
            batch = SimpleNamespace(x=x, C=C, cell_idx=idx)
        for step in range(25):
            if kind == "graph":
                loss = gs(batch)
            else:
                opt.zero_grad(set_to_none=True)
                loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
                opt.scale_loss(loss).backward()
            opt.step()
            losses.append(loss.detach().clone())
            del loss  # (no eager autograd graph may be alive when the capture begins)
        torch.cuda.synchronize()
        ends[kind] = ([l.item() for l in losses], {n: p.detach().clone() for n, p in diff.model.named_parameters()})
    assert ends["eager"][0] == ends["graph"][0]
    assert ends["eager"][0][0] != ends["eager"][0][-1]
    differ = [n for n in ends["eager"][1] if not torch.equal(ends["eager"][1][n], ends["graph"][1][n])]
    assert not differ, differ
    bench.set_mode(diff, "f32")


@pytest.mark.timeout(1500)
def test_benchmark_batch_gradients_match_oracle(monkeypatch):
    """The benchmarked shape itself: B = 6 at 192 x 64 x 48 (per-sample strides, level-1 launches on the ring kernels,
    six samples in the split-K weight gradients and GroupNorm statistics): loss and 18 parameter gradients of p_losses
    in all three modes against the CPU oracle (its fwd + bwd at B = 6 runs once, on the host cores)."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    B = 6
    net, sd = _full_size_problem(seed=2)
    X, Y, Z = 192, 64, 48
    x = torch.randn(B, 4, X, Y, Z, generator=torch.Generator().manual_seed(1234))
    c_local = torch.randn(4, X, Y, Z, generator=torch.Generator().manual_seed(1235))
    noise = torch.randn(B, 4, X, Y, Z, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([3, 250, 499, 17, 120, 380])
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[13:25, 24:40, 0:32] = False
    cell_idx = m.flatten().nonzero().flatten()
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 500)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=500, noise_bcs=True)
    ref_loss.backward()
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    md = SimpleNamespace(cell_idx=cell_idx.to(dev()))
    _check_modes_against_oracle(monkeypatch, diff, (x, t, c_local, noise), md, ref_loss, leaves, WATCHED_FULL_SIZE)


def test_reference_grid_194x50x50_four_levels_forward(monkeypatch):
    """The reference's real grid (194 x 50 x 50: every axis leaves a remainder against the 4 x 8 x 8 brick, every
    level down to 24 x 6 x 6 is ragged) through the full 4-level dim-32 net: forward vs the CPU oracle in the fp32
    and split-precision modes (1e-4) and in bf16 (3e-2)."""
    net, sd = _full_size_problem(seed=3)
    x = torch.randn(1, 4, 194, 50, 50, generator=torch.Generator().manual_seed(5))
    c_local = torch.randn(4, 194, 50, 50, generator=torch.Generator().manual_seed(6))
    t = torch.tensor([77])
    with torch.no_grad():
        ref = O.denoiser(sd, x, t, c_local, timesteps=500)
        net.to(dev())
        y = net(x.to(dev()), t.to(dev()), cond(c_local))
        monkeypatch.setenv("TDX_CONV_IMPL", "split")
        ys = net(x.to(dev()), t.to(dev()), cond(c_local))
        monkeypatch.delenv("TDX_CONV_IMPL")
        net.set_compute_dtype(torch.bfloat16)
        yb = net(x.to(dev()), t.to(dev()), cond(c_local))
    assert rel_l2(y.cpu(), ref) < 1e-4 and rel_l2(ys.cpu(), ref) < 1e-4 and rel_l2(yb.cpu(), ref) < 3e-2


@pytest.mark.timeout(900)
def test_reference_grid_194x50x50_gradients_match_oracle(monkeypatch):
    """the reference's real grid (scripts/grid-embedding.py:69; levels 194x50x50 -> 97x25x25 -> 48x12x12
    -> 24x6x6 -> 12x3x3 via max(int(s/2), 3), ddpm.py:358) through the full 4-level dim-32 net, BACKWARD included: loss and
    the 18 watched parameter gradients of p_losses against the CPU oracle in f32 / f32s (1e-3) and bf16 (0.1), B = 1.  This is
    where the thin-slab data gradient, the ragged weight-gradient bricks and the halo shell run at the real size."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    net, sd = _full_size_problem(seed=3)
    X, Y, Z = 194, 50, 50
    x = torch.randn(1, 4, X, Y, Z, generator=torch.Generator().manual_seed(5))
    c_local = torch.randn(4, X, Y, Z, generator=torch.Generator().manual_seed(6))
    noise = torch.randn(1, 4, X, Y, Z, generator=torch.Generator().manual_seed(7))
    t = torch.tensor([77])
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[40:60, 17:33, 0:25] = False
    cell_idx = m.flatten().nonzero().flatten()
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 500)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=500, noise_bcs=True)
    ref_loss.backward()
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    md = SimpleNamespace(cell_idx=cell_idx.to(dev()))
    _check_modes_against_oracle(monkeypatch, diff, (x, t, c_local, noise), md, ref_loss, leaves, WATCHED_FULL_SIZE)


@pytest.mark.timeout(1500)
def test_reference_grid_194x50x50_benchmark_batch_gradients_match_oracle(monkeypatch):
    """the real grid at the batch `bench.py extra.real_grid` runs it with, B = 6 (the ring-kernel
    eligibility nb * ntn >= 768 depends on the batch, tdx_conv3_ring.hip: B = 1 and B = 6 take different kernels on this
    grid): loss and the 18 watched gradients vs the oracle in f32 / f32s (1e-3) and bf16 (0.1)."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    B = 6
    net, sd = _full_size_problem(seed=4)
    X, Y, Z = 194, 50, 50
    x = torch.randn(B, 4, X, Y, Z, generator=torch.Generator().manual_seed(15))
    c_local = torch.randn(4, X, Y, Z, generator=torch.Generator().manual_seed(16))
    noise = torch.randn(B, 4, X, Y, Z, generator=torch.Generator().manual_seed(17))
    t = torch.tensor([5, 77, 499, 301, 150, 0])
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[40:60, 17:33, 0:25] = False
    cell_idx = m.flatten().nonzero().flatten()
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 500)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=500, noise_bcs=True)
    ref_loss.backward()
    diff = GaussianDiffusion(net, timesteps=500, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    md = SimpleNamespace(cell_idx=cell_idx.to(dev()))
    _check_modes_against_oracle(monkeypatch, diff, (x, t, c_local, noise), md, ref_loss, leaves, WATCHED_FULL_SIZE)


@pytest.mark.timeout(1500)
def test_config3_full_size_graph_sampler_vs_oracle(monkeypatch):
    """VERDICT r4 item 1 -- BASELINE configs[3] at its own size: GaussianDiffusion(T = 1000) on the dim-32, 4-level net at
    192 x 64 x 48, the last three reverse steps (start_from = 3) through the DEFAULT path of p_sample_loop (captured
    hipGraph, in-kernel Philox noise on 589 824-voxel planes, deferred encoder, cached conditioning conv, fused decoder
    tail, ring kernels at B > 1 -- the inference-only routes that exist only at this size) against oracle.p_sample_loop
    (reference ddpm.py:767-816) fed the sampler's own noise stream: B = 2 with trajectory ids (5, 9) in f32 / f32s (1e-4)
    and bf16 (3e-2); then the same three steps at B = 8 (the per-GPU shard of configs[3]) must give, for the shared
    ids, the B = 2 samples: sharding invariance at full size."""
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion
    from turbdiff_amd.sampling import GraphSampler

    torch.manual_seed(11)
    T = 1000
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=T, dim=32,
                         u_net_levels=4, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    X, Y, Z = 192, 64, 48
    gen = torch.Generator().manual_seed(4321)
    xb8 = torch.randn(8, 4, X, Y, Z, generator=gen)
    c_local = torch.randn(4, X, Y, Z, generator=gen)
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[13:25, 24:40, 0:32] = False
    cell_idx = m.flatten().nonzero().flatten()
    ids8 = [5, 9, 1, 2, 3, 4, 6, 7]
    diff = GaussianDiffusion(net, timesteps=T, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    C, ci = cond(c_local), cell_idx.to(dev())
    buf = O.schedule_buffers("log-snr-linear", T)
    ref = None
    for mode, impl, dtype, tol in (("f32", "auto", torch.float32, 1e-4), ("f32s", "split", torch.float32, 1e-4),
                                   ("bf16", "auto", torch.bfloat16, 3e-2), ("fp16", "auto", torch.float16, 4e-3)):
        monkeypatch.setenv("TDX_CONV_IMPL", impl)
        diff.model.set_compute_dtype(dtype)
        out2 = diff.p_sample_loop(xb8[:2].to(dev()), C, ci, start_from=3, seed=77, trajectory_ids=ids8[:2])
        (gs,) = [g for g in diff.graph_samplers().values() if g.B == 2 and g.graph is not None][-1:]
        assert gs.fused_noise, "full-size planes are multiples of four: the update kernel must draw its own noise"
        if ref is None:  # the oracle's three reverse steps at B = 2, on the very noise the sampler drew (~15 s)
            stream = GraphSampler(diff, xb8[:2].to(dev()), C, ci, seed=0, trajectory_ids=ids8[:2], nonce=77,
                                  use_graph=False).noise_stream()
            noises = [next(stream).cpu() for _ in range(1 + 2 * 2)]
            with torch.no_grad():
                ref = O.p_sample_loop(sd, buf, xb8[:2], c_local, cell_idx, noises, timesteps=T, noise_bcs=True, start_from=3)
        assert rel_l2(out2.cpu(), ref) < tol, (mode, rel_l2(out2.cpu(), ref))
        out8 = diff.p_sample_loop(xb8.to(dev()), C, ci, start_from=3, seed=77, trajectory_ids=ids8)
        # same ids, same nonce: the same noise; the arithmetic differs only by kernel selection with the batch
        assert rel_l2(out8[:2], out2) < {"f32": 1e-5, "f32s": 1e-5, "bf16": 2e-2, "fp16": 3e-3}[mode], (mode, rel_l2(out8[:2], out2))
        assert rel_l2(out8[2:].cpu(), out8[:2].repeat(3, 1, 1, 1, 1).cpu()) > 1e-2  # other ids: other samples
        inside = torch.zeros(X * Y * Z, dtype=torch.bool)
        inside[cell_idx] = True
        assert torch.equal(out8.cpu().flatten(-3)[..., ~inside], xb8.flatten(-3)[..., ~inside])
    monkeypatch.delenv("TDX_CONV_IMPL")


@pytest.mark.timeout(600)
def test_config0_48x32x32_two_levels_training_and_sampling_vs_oracle(monkeypatch):
    """VERDICT r3 item 1b -- BASELINE configs[0] as a whole on the GPU: DenoisingModel(dim=32, u_net_levels=2, T=10) at
    48x32x32, (i) loss and EVERY parameter gradient of p_losses vs the oracle in the three modes, (ii) the 10-step
    p_sample_loop with injected noise vs oracle.p_sample_loop (fp32 modes 1e-4 on the sample; bf16 3e-2), for both
    noise_bcs settings."""
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    torch.manual_seed(0)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=32,
                         u_net_levels=2, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    X, Y, Z = 48, 32, 32
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 4, X, Y, Z, generator=gen)
    c_local = torch.randn(4, X, Y, Z, generator=torch.Generator().manual_seed(1235))
    noise = torch.randn(1, 4, X, Y, Z, generator=gen)
    t = torch.tensor([6])
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    m[13:25, 8:24, 0:16] = False
    cell_idx = m.flatten().nonzero().flatten()
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 10)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=10, noise_bcs=True)
    ref_loss.backward()
    diff = GaussianDiffusion(net, timesteps=10, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    md = SimpleNamespace(cell_idx=cell_idx.to(dev()))
    watched = [n for n, p in net.named_parameters() if leaves[n].grad is not None and leaves[n].grad.abs().max() > 1e-7]
    assert len(watched) > 50
    _check_modes_against_oracle(monkeypatch, diff, (x, t, c_local, noise), md, ref_loss, leaves, watched)
    # (ii) the 10-step loop
    sdn = {k: v.detach() for k, v in sd.items()}
    for nb in (True, False):
        noises = [torch.randn(1, 4, X, Y, Z, generator=gen) for _ in range(1 + 9 * (2 if nb else 1))]
        with torch.no_grad():
            ref = O.p_sample_loop(sdn, buf, x, c_local, cell_idx, noises, timesteps=10, noise_bcs=nb)
        diff.noise_bcs = nb
        for mode, impl, dtype, tol in (("f32", "auto", torch.float32, 1e-4), ("f32s", "split", torch.float32, 1e-4),
                                       ("bf16", "auto", torch.bfloat16, 3e-2), ("fp16", "auto", torch.float16, 4e-3)):
            monkeypatch.setenv("TDX_CONV_IMPL", impl)
            diff.model.set_compute_dtype(dtype)
            it = iter([n.to(dev()) for n in noises])
            out = diff.p_sample_loop(x.to(dev()), cond(c_local), md.cell_idx, noise_fn=lambda like: next(it))
            assert next(it, None) is None
            assert rel_l2(out.cpu(), ref) < tol, (mode, nb, rel_l2(out.cpu(), ref))
        monkeypatch.delenv("TDX_CONV_IMPL")
        diff.model.set_compute_dtype(torch.float32)


def test_split_precision_training_step_vs_oracle(monkeypatch):
    """dim 32, 3 levels, 48x32x24, B = 2: loss and every parameter gradient of p_losses with
    TDX_CONV_IMPL=split (fp32 tensors, bf16 hi + lo MFMA convs) against the CPU oracle."""
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    torch.manual_seed(1)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=32,
                         u_net_levels=3, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 48, 32, 24, generator=gen)
    c_local = torch.randn(4, 48, 32, 24, generator=gen)
    noise = torch.randn(2, 4, 48, 32, 24, generator=gen)
    t = torch.tensor([3, 8])
    inside = torch.zeros(48, 32, 24, dtype=torch.bool)
    inside[1:-1, 1:-1, 1:-1] = True
    cell_idx = inside.flatten().nonzero().flatten()
    # oracle with autograd on leaf copies of the parameters
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    buf = O.schedule_buffers("log-snr-linear", 10)
    ref_loss, _ = O.p_losses(leaves, buf, x, t, c_local, cell_idx, noise, timesteps=10, noise_bcs=True)
    ref_loss.backward()
    monkeypatch.setenv("TDX_CONV_IMPL", "split")
    diff = GaussianDiffusion(net, timesteps=10, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    loss, _ = diff.p_losses(x.to(dev()), t.to(dev()), cond(c_local), SimpleNamespace(cell_idx=cell_idx.to(dev())), None,
                            noise=noise.to(dev()))
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    for name, p in net.named_parameters():
        assert_grad_close(name, p.grad.cpu(), leaves[name].grad, 1e-3)


def test_sampling_is_linear_in_boundary_values_property():
    """size-independent property at the full grid: with eps_theta == 0 the fused reverse
    step is affine in (x_t, z); checked on the 192x64x48 grid through the C ABI."""
    from turbdiff_amd import ops, schedules

    d = dev()
    T = 500
    sched = schedules.pack_step_tables(schedules.diffusion_tables("log-snr-linear", T)).to(d)
    shape = (2, 4, 192, 64, 48)
    V = 192 * 64 * 48
    g = torch.Generator(device=d).manual_seed(0)
    a, b, z, xb = (torch.randn(shape, device=d, generator=g) for _ in range(4))
    zero = torch.zeros(shape, device=d)
    mask = torch.ones(V, dtype=torch.uint8, device=d)
    tt = torch.tensor([250], device=d)
    step = lambda x, zz: ops.p_sample_step(x, zero, zz, zero, xb, mask, sched, T, tt, True, False)
    lhs = step(a + b, z)
    rhs = step(a, z) + step(b, zero)
    assert rel_l2(lhs, rhs) < 1e-6


@pytest.mark.parametrize("nb", [True, False])
def test_graph_sampler_equals_eager_loop(golden, nb):
    """The hipGraph-captured sampler (device-side t, Philox noise) reproduces the eager
    p_sample_loop fed with the very same noise tensors."""
    from turbdiff_amd.sampling import GraphSampler

    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=nb)
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())
    gs = GraphSampler(diff, x_bcs, C, cidx, seed=42, trajectory_ids=[5, 9])
    out_graph = gs.sample()
    stream = gs.noise_stream()
    out_eager = diff.p_sample_loop(x_bcs, C, cidx, noise_fn=lambda like: next(stream))
    assert rel_l2(out_graph, out_eager) < 1e-5
    # replaying gives the same trajectory again (offset / t reset on device)
    # (GroupNorm statistics merge per-block partials with f64 atomics, so runs agree to
    # rounding, not bitwise)
    assert rel_l2(gs.sample(), out_graph) < 1e-5
    # sharding invariance: trajectory 9 alone, on its own "rank", gives the same sample
    solo = GraphSampler(diff, x_bcs[1:], C, cidx, seed=42, trajectory_ids=[9], use_graph=False).sample()
    assert rel_l2(solo[0], out_graph[1]) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sampling_is_bit_reproducible_under_the_deterministic_switch(golden, dtype, monkeypatch):
    """TDX_DETERMINISTIC=1: the captured sampler gives the SAME BITS when replayed (the GroupNorm statistics of the forward merge
    through per-block tables in block order instead of f64 atomics; the first conv's partial-forward entry included), and the
    eager loop on the same noise gives those bits too."""
    from turbdiff_amd.sampling import GraphSampler

    monkeypatch.setenv("TDX_DETERMINISTIC", "1")
    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=True, dtype=dtype)
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())
    gs = GraphSampler(diff, x_bcs, C, cidx, seed=42, trajectory_ids=[5, 9])
    first = gs.sample().clone()
    assert torch.isfinite(first).all()
    for _ in range(3):
        assert torch.equal(gs.sample(), first)
    stream = gs.noise_stream()
    eager = diff.p_sample_loop(x_bcs, C, cidx, noise_fn=lambda like: next(stream))
    assert torch.equal(eager, first)


@pytest.mark.parametrize("nb", [True, False])
def test_p_sample_loop_default_path_is_the_graph_sampler(golden, nb):
    """`GaussianDiffusion.p_sample_loop` without injected noise -- what `DiffusionTrainer.sample`,
    `tools/eval_ckpt.py` and a `dropin` user call (reference diffusion.py:152-158, ddpm.py:767-816) -- runs the captured
    reverse step.  (a) It equals the eager loop fed with the very noise it drew (seed / trajectory ids given);
    (b) a second call with ANOTHER geometry and batch of the same shape re-uses the graph (no re-capture) and is again
    equal to its eager twin; (c) `start_from` works through the same graph; (d) with the nonce drawn from torch's global
    generator, `torch.manual_seed` makes two calls agree and two different seeds differ; (e) pbar=True runs."""
    from turbdiff_amd.models import ddpm as D
    from turbdiff_amd.sampling import GraphSampler

    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=nb)
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())

    def eager_twin(xb, Cc, ci, nonce, ids, start=None):
        ref = GraphSampler(diff, xb, Cc, ci, seed=0, trajectory_ids=ids, nonce=nonce, use_graph=False)
        stream = ref.noise_stream()
        return diff.p_sample_loop(xb, Cc, ci, start_from=start, noise_fn=lambda like: next(stream))

    out = diff.p_sample_loop(x_bcs, C, cidx, seed=123, trajectory_ids=[5, 9])
    (gs,) = diff.graph_samplers().values()
    assert gs.graph is not None, "the default path did not capture a graph"
    assert rel_l2(out, eager_twin(x_bcs, C, cidx, 123, [5, 9])) < 1e-5
    first_graph = gs.graph
    # (b) another geometry: shifted obstacle, other boundary values and conditioning
    X, Y, Z = x_bcs.shape[-3:]
    m = torch.zeros(X, Y, Z, dtype=torch.bool)
    m[1:-1, 1:-1, 2:-1] = True
    m[2:4, 3:6, 2:5] = False
    cidx2 = m.flatten().nonzero().flatten().to(dev())
    gen = torch.Generator().manual_seed(77)
    x2 = torch.randn(x_bcs.shape, generator=gen).to(dev())
    C2 = cond(torch.randn(g["c_local"].shape, generator=gen))
    out2 = diff.p_sample_loop(x2, C2, cidx2, seed=7, trajectory_ids=[0, 1])
    assert list(diff.graph_samplers().values()) == [gs] and gs.graph is first_graph, "same shapes: the captured graph must be re-used"
    assert rel_l2(out2, eager_twin(x2, C2, cidx2, 7, [0, 1])) < 1e-5
    inside = torch.zeros(X * Y * Z, dtype=torch.bool)
    inside[cidx2.cpu()] = True
    assert torch.equal(out2.cpu().flatten(-3)[..., ~inside], x2.cpu().flatten(-3)[..., ~inside])
    # (c) start_from
    out3 = diff.p_sample_loop(x2, C2, cidx2, seed=9, start_from=5)
    assert rel_l2(out3, eager_twin(x2, C2, cidx2, 9, [0, 1], start=5)) < 1e-5
    # (d) reproducible under torch.manual_seed, like the reference's torch.randn_like draws
    torch.manual_seed(31)
    a = diff.p_sample_loop(x_bcs, C, cidx)
    torch.manual_seed(31)
    b = diff.p_sample_loop(x_bcs, C, cidx, pbar=True)
    torch.manual_seed(32)
    c = diff.p_sample_loop(x_bcs, C, cidx)
    assert rel_l2(a, b) < 1e-5 and rel_l2(a, c) > 1e-2
    assert gs.graph is first_graph


def test_p_sample_loop_eager_switch(golden, monkeypatch):
    """TDX_GRAPH_SAMPLER=0 (models.ddpm.GRAPH_SAMPLER): the default path is the eager loop with torch.randn_like."""
    from turbdiff_amd.models import ddpm as D

    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=True)
    monkeypatch.setattr(D, "GRAPH_SAMPLER", False)
    torch.manual_seed(3)
    out = diff.p_sample_loop(g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev()))
    assert not diff.graph_samplers() and torch.isfinite(out).all()


def test_graph_samplers_die_with_their_diffusion_and_are_kept_per_shape(golden):
    """(a) a sampled diffusion that is dropped is collected together with its captured sampler (graph, private
    pool, buffers) -- the former WeakKeyDictionary kept both alive for the whole process; (b) calls that alternate
    between shapes keep one captured sampler each (LRU of MAX_GRAPH_SAMPLERS) on ONE capture stream, and flipping back
    does not re-capture; (c) copy.deepcopy of a sampled diffusion works and starts without samplers; (d) run_steps on a
    finished sampler is a no-op (no capture at t = -1)."""
    import copy
    import gc
    import weakref

    from turbdiff_amd.models import ddpm as D

    g = golden("sample_cfg1")
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())
    diff = build_cfg1(golden, noise_bcs=True)
    diff.p_sample_loop(x_bcs, C, cidx, seed=1)
    (gs2,) = diff.graph_samplers().values()
    graph2 = gs2.graph
    diff.p_sample_loop(x_bcs[:1], C, cidx, seed=1)  # another shape: a second sampler, the first one stays
    assert len(diff.graph_samplers()) == 2
    gs1 = list(diff.graph_samplers().values())[-1]
    assert gs1 is not gs2 and gs1._capture_stream is gs2._capture_stream
    diff.p_sample_loop(x_bcs, C, cidx, seed=2)      # back to the first shape: same sampler, same graph
    assert list(diff.graph_samplers().values())[-1] is gs2 and gs2.graph is graph2
    for k in range(D.MAX_GRAPH_SAMPLERS + 1):        # more shapes than the cache holds: the oldest go
        xk = x_bcs[:1, :, : x_bcs.shape[2] - 2 * (k + 1)].contiguous()
        X = xk.shape[2]
        ck = {key: v[:, :X].contiguous() for key, v in C.items()}
        m = torch.zeros(xk.shape[-3:], dtype=torch.bool)
        m[1:-1, 1:-1, 1:-1] = True
        diff.p_sample_loop(xk, ck, m.flatten().nonzero().flatten().to(dev()), seed=3)
    assert len(diff.graph_samplers()) == D.MAX_GRAPH_SAMPLERS and gs2 not in diff.graph_samplers().values()
    # (d)
    last = list(diff.graph_samplers().values())[-1]
    assert last.steps_left == 0
    before = last.x_t.clone()
    last.graph = None
    assert last.run_steps(5) is last.x_t and last.graph is None and torch.equal(before, last.x_t)
    # (c)
    twin = copy.deepcopy(diff)
    assert not twin.graph_samplers() and torch.isfinite(twin.p_sample_loop(x_bcs, C, cidx, seed=1)).all()
    del twin
    # (a)
    del gs1, gs2, graph2, last
    probe = build_cfg1(golden, noise_bcs=True)
    probe.p_sample_loop(x_bcs, C, cidx, seed=1)
    sampler = next(iter(probe.graph_samplers().values()))
    wd, ws, wx = weakref.ref(probe), weakref.ref(sampler), weakref.ref(sampler.x_t)
    del probe, sampler
    gc.collect()
    torch.cuda.synchronize()
    assert wd() is None and ws() is None and wx() is None, "a dropped diffusion must not be kept alive by its sampler cache"


def test_graph_sampler_recaptures_after_a_weight_update(golden):
    """The captured graph has the packed weight operands' addresses baked in.  After the weights change (optimiser
    step, load_state_dict, here an in-place update) the sampler must not replay stale operands: it re-captures, and
    the result equals the eager loop on the NEW weights."""
    from turbdiff_amd.sampling import GraphSampler

    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=True)
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())
    gs = GraphSampler(diff, x_bcs, C, cidx, seed=7)
    before = gs.sample()
    first_graph = gs.graph
    with torch.no_grad():
        for p in diff.model.parameters():
            p.mul_(1.05)  # bumps Tensor._version, as an optimiser step does
    after = gs.sample()
    assert gs.graph is not first_graph, "the sampler kept replaying the graph captured on the old weights"
    stream = gs.noise_stream()
    eager = diff.p_sample_loop(x_bcs, C, cidx, noise_fn=lambda like: next(stream))
    assert rel_l2(after, eager) < 1e-5 and rel_l2(after, before) > 1e-3
    # a parameter REPLACED by another tensor (no version bump on the old one) is a change too
    second_graph = gs.graph
    net = diff.model
    net.decode[1].weight = torch.nn.Parameter(net.decode[1].weight.detach() * 0.5)
    swapped = gs.sample()
    assert gs.graph is not second_graph
    stream = gs.noise_stream()
    eager = diff.p_sample_loop(x_bcs, C, cidx, noise_fn=lambda like: next(stream))
    assert rel_l2(swapped, eager) < 1e-5 and rel_l2(swapped, after) > 1e-3


def test_graph_sampler_on_a_grid_whose_planes_are_not_multiples_of_four():
    """13 x 7 x 9 = 819 voxels per plane: the in-kernel noise draw (16-B groups per plane) does not apply, the sampler falls
    back to drawing z / z2 into tensors; graph and eager loops still agree, and so do a fused-noise and a separate-noise
    sampler on a grid where both work (same Philox counters)."""
    from turbdiff_amd import sampling
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion
    from turbdiff_amd.sampling import GraphSampler

    torch.manual_seed(2)
    for grid, fused in (((13, 7, 9), False), ((12, 8, 10), True)):
        net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=6, dim=16,
                             u_net_levels=2, norm_type="group")
        diff = GaussianDiffusion(net, timesteps=6, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
        V = grid[0] * grid[1] * grid[2]
        x_bcs = torch.randn(2, 4, *grid, generator=torch.Generator().manual_seed(4)).to(dev())
        C = cond(torch.randn(4, *grid, generator=torch.Generator().manual_seed(5)))
        cidx = torch.arange(0, V, 3, device=dev())
        gs = GraphSampler(diff, x_bcs, C, cidx, seed=3, trajectory_ids=[0, 7])
        assert gs.fused_noise == fused and (gs.z is None) == fused
        out = gs.sample()
        stream = gs.noise_stream()
        eager = diff.p_sample_loop(x_bcs, C, cidx, noise_fn=lambda like: next(stream))
        assert torch.isfinite(out).all() and rel_l2(out, eager) < 1e-5
        if fused:
            import unittest.mock as um
            with um.patch.object(sampling, "FUSED_STEP_NOISE", False):
                plain = GraphSampler(diff, x_bcs, C, cidx, seed=3, trajectory_ids=[0, 7])
            assert not plain.fused_noise and rel_l2(plain.sample(), out) < 1e-5


def test_sampler_refreshes_its_cached_conditioning_conv_after_a_weight_update(monkeypatch):
    """With the first conv NOT composed with the encoders (models.ddpm.COMPOSE_FIRST_CONV = False) the sampler keeps
    encode_local(C) and the conditioning half of the first conv (first_conv_partial): both are functions of the weights
    and must follow a weight update, into the same tensors (the captured graph reads those addresses)."""
    from turbdiff_amd.models import ddpm as D
    from turbdiff_amd.sampling import GraphSampler

    monkeypatch.setattr(D, "COMPOSE_FIRST_CONV", False)
    torch.manual_seed(1)
    grid = (16, 8, 8)
    net = D.DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=6, dim=32,
                           u_net_levels=2, norm_type="group")
    diff = D.GaussianDiffusion(net, timesteps=6, beta_schedule="log-snr-linear", noise_bcs=True).to(dev())
    net.set_compute_dtype(torch.bfloat16)
    V = grid[0] * grid[1] * grid[2]
    x_bcs = torch.randn(2, 4, *grid, generator=torch.Generator().manual_seed(4)).to(dev())
    C = cond(torch.randn(4, *grid, generator=torch.Generator().manual_seed(5)))
    cidx = torch.arange(0, V, 3, device=dev())
    gs = GraphSampler(diff, x_bcs, C, cidx, seed=3)
    part = getattr(gs.enc, "first_conv_partial", None)
    assert part is not None, "this configuration should use the cached conditioning conv"
    held, old_part, old_enc = part[1], part[1].clone(), gs.enc.clone()
    gs.sample()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.25)
    after = gs.sample()
    assert gs.enc.first_conv_partial[1] is held  # same tensor (the captured graph reads its address) ...
    with torch.no_grad():
        fresh = net.encode_local(C)
    # ... with the new weights' contents: equal to a fresh evaluation (bf16 rounding), far from the old ones
    assert rel_l2(gs.enc.float(), fresh.float()) < 1e-2 and rel_l2(held.float(), fresh.first_conv_partial[1].float()) < 1e-2
    assert rel_l2(gs.enc.float(), old_enc.float()) > 0.1 and rel_l2(held.float(), old_part.float()) > 0.1
    assert torch.isfinite(after).all()


def test_conditioning_table_row_equals_the_time_mlp(golden):
    """DenoisingModel.conditioning_table: row t is the conditioning vector the model computes for timestep t (the sampler
    looks it up instead of running the time MLP every reverse step); a forward with cond = those rows equals the plain
    forward; the table is refused (None) when the vector depends on more than t; the sampler refreshes its copy after a
    weight update (same tensor, new values)."""
    from turbdiff_amd.sampling import GraphSampler

    g = golden("sample_cfg1")
    diff = build_cfg1(golden, noise_bcs=True)
    net = diff.model
    x_bcs, C, cidx = g["x_bcs"].to(dev()), cond(g["c_local"]), g["cell_idx"].to(dev())
    T = diff.num_timesteps
    with torch.no_grad():
        tab = net.conditioning_table(C, T)
        assert tab is not None and tab.shape[0] == T
        t = torch.tensor([0, T - 1], device=dev())
        direct = net.conditioning_vector(t, C, 2)
        assert rel_l2(tab[t], direct) < 1e-6
        x = torch.randn(2, *x_bcs.shape[1:], generator=torch.Generator().manual_seed(3)).to(dev())
        assert rel_l2(net(x, t, C, cond=tab[t]), net(x, t, C)) < 1e-5
        net.with_geometry_embedding = True  # the vector would then depend on the geometry too
        assert net.conditioning_table(C, T) is None
        net.with_geometry_embedding = False
    gs = GraphSampler(diff, x_bcs, C, cidx, seed=7)
    assert gs.c_table is not None and rel_l2(gs.c_table, tab) < 1e-6
    held = gs.c_table
    gs.run_steps(1)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.05)
    gs.run_steps(1)
    with torch.no_grad():
        assert gs.c_table is held and rel_l2(gs.c_table, net.conditioning_table(C, T)) < 1e-6 and rel_l2(held, tab) > 1e-3


def test_unfused_block_composition_matches_golden(golden, monkeypatch):
    """models.ddpm.FUSE_BLOCKS = False path (one autograd node per operator) -- same kernels, cross-check of the
    hand-written ResnetBlock backward used by default."""
    import turbdiff_amd.models.ddpm as D

    monkeypatch.setattr(D, "FUSE_BLOCKS", False)
    g = golden("model_cfg1")
    diff = build_cfg1(golden, noise_bcs=True)
    loss, _ = diff.p_losses(g["x"].to(dev()), g["t"].to(dev()), cond(g["c_local"]),
                            SimpleNamespace(cell_idx=g["cell_idx"].to(dev())), None, noise=g["loss_nb1/noise"].to(dev()))
    loss.backward()
    assert abs(loss.item() - g["loss_nb1/loss"].item()) < 1e-4 * abs(g["loss_nb1/loss"].item())
    for name, p in diff.model.named_parameters():
        assert_grad_close(name, p.grad.cpu(), g[f"loss_nb1/grad/{name}"], 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fused_resnet_block_golden(golden, dtype):
    """ops.resnet_block (one autograd node, hand-written backward) against the reference's
    ResnetBlock vectors: 1x1-projected skip (8 -> 16) and identity skip (16 -> 16)."""
    from turbdiff_amd import ops

    g = golden("ops")
    d = dev()
    nvc = lambda x: x.permute(0, 2, 3, 4, 1).contiguous()
    ncv = lambda x: x.permute(0, 4, 1, 2, 3).contiguous()
    tol, gtol = {torch.float32: (1e-4, 1e-3), torch.bfloat16: (3e-2, 8e-2), torch.float16: (4e-3, 1e-2)}[dtype]
    for tag in ["resnet_proj", "resnet_id"]:
        sd = {k: v.to(d).requires_grad_() for k, v in g.sub(f"{tag}/sd/").items()}
        x = nvc(g[f"{tag}/x"]).to(d).to(dtype).requires_grad_()
        c = g[f"{tag}/c"].to(d).requires_grad_()
        film = torch.nn.functional.linear(c, sd["project_onto_scale_shift.weight"], sd["project_onto_scale_shift.bias"])
        Cout = film.shape[1] // 2
        skip = (sd["conv.weight"], sd["conv.bias"]) if "conv.weight" in sd else None
        y = ops.resnet_block(x, None, film[:, :Cout], film[:, Cout:], (sd["block1.conv.weight"], sd["block1.conv.bias"]),
                             (sd["block1.norm.weight"], sd["block1.norm.bias"]), (sd["block2.conv.weight"], sd["block2.conv.bias"]),
                             (sd["block2.norm.weight"], sd["block2.norm.bias"]), skip, 8)
        y.backward(nvc(g[f"{tag}/gy"]).to(d).to(dtype))
        assert rel_l2(ncv(y.float().cpu()), g[f"{tag}/y"]) < tol, tag
        assert rel_l2(ncv(x.grad.float().cpu()), g[f"{tag}/gx"]) < gtol, tag
        assert rel_l2(c.grad.cpu(), g[f"{tag}/gc"]) < gtol, tag
        for k, v in sd.items():
            assert_grad_close(f"{tag}/{k}", v.grad.cpu(), g[f"{tag}/grad/{k}"], gtol)


def test_two_trainers_in_different_modes_keep_their_own_conv_arithmetic(golden, monkeypatch):
    """an f32 trainer built after an f32s one used to reset the f32s model to IEEE-fp32 convs (process-global
    switch).  The f32s model's output must not change when another trainer is built, must differ from the f32 model's on
    the same weights (different arithmetic), and its backward -- which runs outside the forward's scope -- must work on the
    split-packed operands."""
    from turbdiff_amd.training import DiffusionTrainer

    monkeypatch.delenv("TDX_CONV_IMPL", raising=False)
    g = golden("model_cfg1")
    d = dev()
    small = {**DiffusionTrainer.SHIPPED_CONFIG, "dim": 8, "timesteps": 10}
    build = lambda mode: DiffusionTrainer(**small, u_net_levels=2, max_train_steps=5, compute_mode=mode).to(d)
    a = build("f32s")
    a.model.model.load_state_dict(g.sub("sd/"))
    x, t = g["x"].to(d), g["t"].to(d)
    cl = cond(torch.randn(a.model.model.c_local_features, *g["x"].shape[-3:], generator=torch.Generator().manual_seed(5)).to(d))
    with torch.no_grad():
        ya = a.model.model(x, t, cl)
    b = build("f32")
    b.model.model.load_state_dict(g.sub("sd/"))
    with torch.no_grad():
        yb = b.model.model(x, t, cl)
        ya2 = a.model.model(x, t, cl)
    assert torch.equal(ya, ya2), "building a second trainer changed the first one's arithmetic"
    assert not torch.equal(ya, yb) and rel_l2(ya.cpu(), yb.cpu()) < 1e-4
    out = a.model.model(x, t, cl)
    out.square().mean().backward()  # data / weight gradients on the split-packed operands, outside the forward's scope
    gb = b.model.model(x, t, cl)
    gb.square().mean().backward()
    for (n, pa), (_, pb) in zip(a.model.model.named_parameters(), b.model.model.named_parameters()):
        # (a conv bias in front of a normalisation has a mathematically zero gradient: rounding noise in both modes)
        if pa.grad is not None and pb.grad is not None and pb.grad.abs().max() > 1e-6:
            assert rel_l2(pa.grad.cpu(), pb.grad.cpu()) < 2e-3, n


def test_trainer_training_step_and_sample(golden):
    """DiffusionTrainer: normalisation + learned cell-type embedding + GaussianDiffusion, checked
    against the oracle fed with the same normalised input / embedded conditioning."""
    from turbdiff_amd.training import DiffusionTrainer

    g = golden("model_cfg1")
    d = dev()
    torch.manual_seed(3)
    task = DiffusionTrainer(**{**DiffusionTrainer.SHIPPED_CONFIG, "dim": 8, "timesteps": 10}, u_net_levels=2, max_train_steps=20).to(d)
    task.model.model.load_state_dict(g.sub("sd/"))
    X, Y, Z = g["x"].shape[-3:]
    gen = torch.Generator().manual_seed(11)
    cell_types = torch.randint(0, 6, (X, Y, Z), generator=gen)
    mean, std = torch.tensor([0.3, -0.1, 0.2, 1.0]), torch.tensor([2.0, 1.5, 0.7, 3.0])
    raw = g["x"] * std.view(4, 1, 1, 1) + mean.view(4, 1, 1, 1)
    batch = SimpleNamespace(x=raw.to(d), cell_idx=g["cell_idx"].to(d), cell_types=cell_types.to(d), mean=mean.to(d), std=std.to(d))
    x_n, C = task._model_input(batch)
    assert rel_l2(x_n.cpu(), g["x"]) < 1e-6
    emb = task.cell_type_embedding.embedding.weight.detach().cpu()
    c_local = emb[cell_types].movedim(-1, 0)
    # loss with injected (t, noise) equals the oracle's
    noise = g["loss_nb1/noise"].to(d)
    loss, _ = task.model.p_losses(x_n, g["t"].to(d), C, SimpleNamespace(cell_idx=batch.cell_idx), None, noise=noise)
    buf = O.schedule_buffers("log-snr-linear", 10)
    ref, _ = O.p_losses(g.sub("sd/"), buf, g["x"], g["t"], c_local, g["cell_idx"], g["loss_nb1/noise"], timesteps=10, noise_bcs=True)
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    loss.backward()
    assert task.cell_type_embedding.embedding.weight.grad is not None  # conditioning is trained
    # two optimiser steps run and change the parameters; loss stays finite
    before = task.model.model.encode_x.weight.detach().clone()
    for _ in range(2):
        l = task.fit_step(batch)
        assert torch.isfinite(l)
    assert not torch.equal(before, task.model.model.encode_x.weight.detach())
    # sample: denormalised output keeps the data values outside the domain (ddpm.py:814 + diffusion.py:157)
    out = task.sample(batch)
    inside = torch.zeros(X * Y * Z, dtype=torch.bool)
    inside[g["cell_idx"]] = True
    assert rel_l2(out.cpu().flatten(-3)[..., ~inside], raw.flatten(-3)[..., ~inside]) < 1e-5
    assert task.measure_sample_time(batch) > 0


def _dense_batch(g, d, seed=11):
    X, Y, Z = g["x"].shape[-3:]
    cell_types = torch.randint(0, 6, (X, Y, Z), generator=torch.Generator().manual_seed(seed))
    mean, std = torch.tensor([0.3, -0.1, 0.2, 1.0]), torch.tensor([2.0, 1.5, 0.7, 3.0])
    raw = g["x"] * std.view(4, 1, 1, 1) + mean.view(4, 1, 1, 1)
    return SimpleNamespace(x=raw.to(d), cell_idx=g["cell_idx"].to(d), cell_types=cell_types.to(d), mean=mean.to(d), std=std.to(d))


def test_fp16_trainer_tracks_the_f32s_trainer_and_recovers_from_overflow(golden):
    """compute_mode="fp16" (round 6): fp16 tensors + fp16 MFMA operands, training under ClipRAdam's power-of-two loss scale.
    (i) Eight optimiser steps from the same weights, the same draws of (t, noise) per step: the fp16 trainer's losses follow
    the split-precision fp32 trainer's within 1 % and its parameters end within 2 % of that trainer's total movement; no step
    is skipped, the scale is the one chosen from the first batch.  (ii) A scale that overflows fp16 (2^40 x the seed)
    skips steps WITHOUT touching parameters or moments, halves its way down and resumes training by itself.  (iii) The
    captured step (enable_graph_step) reads the scale from a device scalar: finite losses, moving parameters."""
    from turbdiff_amd.training import DiffusionTrainer

    g = golden("model_cfg1")
    d = dev()
    batch = _dense_batch(g, d)
    cfg = {**DiffusionTrainer.SHIPPED_CONFIG, "dim": 8, "timesteps": 10}

    def make(mode):
        torch.manual_seed(3)
        task = DiffusionTrainer(**cfg, u_net_levels=2, max_train_steps=20, compute_mode=mode).to(d)
        task.model.model.load_state_dict(g.sub("sd/"))
        return task

    ref, half = make("f32s"), make("fp16")
    start = {k: v.detach().clone() for k, v in ref.model.model.named_parameters()}
    for i in range(8):
        torch.manual_seed(100 + i)
        la = ref.fit_step(batch).item()
        torch.manual_seed(100 + i)
        lb = half.fit_step(batch).item()
        assert abs(la - lb) < 1e-2 * abs(la), (i, la, lb)
    half._opt.settle()
    n_loss = int(batch.x.shape[0]) * 4 * int(batch.cell_idx.numel())
    assert half._opt.skipped_steps == 0 and half._opt.loss_scale == DiffusionTrainer.initial_loss_scale(n_loss)
    pa, pb = dict(ref.model.model.named_parameters()), dict(half.model.model.named_parameters())
    moved = sum(float((pa[k].detach() - start[k]).norm() ** 2) for k in pa) ** 0.5
    apart = sum(float((pa[k].detach() - pb[k].detach()).norm() ** 2) for k in pa) ** 0.5
    assert apart < 2e-2 * moved, (apart, moved)
    # (ii) overflow
    half._opt.loss_scale = 2.0**40
    frozen = {k: v.detach().clone() for k, v in pb.items()}
    m0 = half._opt.state[pb["encode_x.weight"]]["exp_avg"].clone()
    for _ in range(3):
        assert torch.isfinite(half.fit_step(batch))  # the loss itself is computed in fp32 from finite activations
    assert all(torch.equal(frozen[k], pb[k].detach()) for k in frozen), "overflowing steps must leave the parameters alone"
    assert torch.equal(m0, half._opt.state[pb["encode_x.weight"]]["exp_avg"])
    for _ in range(40):
        half.fit_step(batch)
    half._opt.settle()
    assert half._opt.skipped_steps >= 10 and half._opt.loss_scale < 2.0**30
    assert not torch.equal(frozen["encode_x.weight"], pb["encode_x.weight"].detach()), "training must resume at a lower scale"
    steps = float(half._opt.state[pb["encode_x.weight"]]["step"])
    assert steps == 8 + 43 - half._opt.skipped_steps, (steps, half._opt.skipped_steps)  # skipped steps are not counted
    # (iii) the captured step
    graphed = make("fp16")
    graphed.enable_graph_step()
    w0 = graphed.model.model.encode_x.weight.detach().clone()
    losses = [graphed.fit_step(batch) for _ in range(4)]
    assert all(torch.isfinite(l) for l in losses) and len({l.data_ptr() for l in losses}) == 4  # fresh tensors
    graphed._opt.settle()
    assert graphed._opt.skipped_steps == 0 and not torch.equal(w0, graphed.model.model.encode_x.weight.detach())


@pytest.mark.parametrize("grid,levels", [((50, 26, 18), 3), ((13, 7, 6), 2), ((97, 25, 25), 2)])
def test_odd_grids_forward_and_grads_vs_oracle(grid, levels, monkeypatch):
    """Grids that do not divide the brick sizes (the reference's real data is 194x50x50 -> 97x25x25
    -> ...; resampling uses max(int(s/2), 3), ddpm.py:358): HIP model vs the CPU oracle, fp32 and
    bf16, forward and parameter gradients."""
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(5)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=50, dim=16,
                         u_net_levels=levels, norm_type="group")
    sd = {k: v.clone().requires_grad_() for k, v in net.state_dict().items()}
    gx = torch.Generator().manual_seed(9)
    x = torch.randn(2, 4, *grid, generator=gx)
    c_local = torch.randn(4, *grid, generator=gx)
    t = torch.tensor([3, 41])
    gy = torch.randn(2, 4, *grid, generator=gx)
    ref = O.denoiser(sd, x, t, c_local, timesteps=50)
    ref.backward(gy)
    net.to(dev())
    # "split": fp32 tensors with split-precision convs where the channel counts allow (dim 16: a mix of split,
    # fp32-MFMA and vector-ALU layers, each with its own packed weight layout)
    for dtype, tol, gtol, impl in [(torch.float32, 1e-4, 2e-3, "auto"), (torch.float32, 1e-4, 2e-3, "split"),
                                   (torch.bfloat16, 3e-2, 0.15, "auto")]:
        monkeypatch.setenv("TDX_CONV_IMPL", impl)
        net.set_compute_dtype(dtype)
        net.zero_grad(set_to_none=True)
        y = net(x.to(dev()), t.to(dev()), cond(c_local))
        y.backward(gy.to(dev()))
        assert rel_l2(y.cpu(), ref) < tol, (dtype, rel_l2(y.cpu(), ref))
        for name, p in net.named_parameters():
            assert_grad_close(name, p.grad.cpu(), sd[name].grad, gtol, noise_floor=1e-5)


def test_local_attention_golden(golden):
    """LocalAttention (reference ddpm.py:232-283) on a grid that needs padding: the windowed
    batches go through the HIP attention kernel (N = 8 keys per window)."""
    from turbdiff_amd.models.ddpm import LocalAttention

    g = golden("ops")
    la = LocalAttention(16, window_size=2, heads=4, dim_head=32)
    la.load_state_dict(g.sub("local_attn/sd/"), strict=True)
    la.to(dev())
    with torch.no_grad():
        y = la(g["local_attn/x"].to(dev()))
    assert rel_l2(y.cpu(), g["local_attn/y"]) < 1e-5


def test_cached_conditioning_conv_matches_plain_forward():
    """Sampling-time shortcut: encode_local() precomputes the conditioning half of the first U-Net conv
    (tdx_conv3_fwd_partial continues from it).  Against the plain forward (same kernels, full conv) and
    against the CPU oracle, bf16, B = 3, a grid with ragged bricks."""
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(0)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                         u_net_levels=2, norm_type="group")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(1))
    c_local = torch.randn(4, 26, 12, 17, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([3, 250, 499])
    net.to(dev()).set_compute_dtype(torch.bfloat16)
    C = cond(c_local)
    with torch.no_grad():
        ref = O.denoiser(sd, x, t, c_local, timesteps=500)
        plain = net(x.to(dev()), t.to(dev()), C)
        enc = net.encode_local(C)
        assert getattr(enc, "first_conv_partial", None) is not None and enc.first_conv_partial[0] == 32
        cached = net(x.to(dev()), t.to(dev()), C, encoded_local=enc)
    assert rel_l2(cached.cpu(), plain.cpu()) < 1e-2
    assert rel_l2(plain.cpu(), ref) < 3e-2 and rel_l2(cached.cpu(), ref) < 3e-2
    # with autograd on, the shortcut is not taken (the backward needs the full conv)
    enc2 = net.encode_local(C)
    assert getattr(enc2, "first_conv_partial", None) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_deferred_encoder_output_is_bit_identical(dtype, monkeypatch):
    """The encoder output feeds only the first block's identity skip once the first conv is composed with the encoders;
    it is then evaluated inside that block's tail kernel (tdx_gn_apply_encoded, ops.encode_deferred) instead of being
    written and read back.  Same output and same parameter gradients, bit for bit, as with models.ddpm.DEFER_ENCODE = False -- in the
    forward (no-grad: what sampling runs) and through the backward (the stand-in tensor carries the skip's gradient to
    the encoders)."""
    from turbdiff_amd import ops as ops_mod
    from turbdiff_amd.models import ddpm as D

    torch.manual_seed(0)
    net = D.DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                           u_net_levels=2, norm_type="group")
    x = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(1)).to(dev())
    c_local = torch.randn(4, 26, 12, 17, generator=torch.Generator().manual_seed(2)).to(dev())
    t = torch.tensor([3, 250, 499]).to(dev())
    net.to(dev()).set_compute_dtype(dtype)
    C = cond(c_local)
    gy = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(3)).to(dev())

    def run(defer):
        monkeypatch.setattr(D, "DEFER_ENCODE", defer)
        calls = []
        orig = ops_mod.encode_deferred
        monkeypatch.setattr(ops_mod, "encode_deferred", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        with torch.no_grad():
            y0 = net(x, t, C).clone()
        net.zero_grad(set_to_none=True)
        y = net(x, t, C)
        (y * gy).sum().backward()
        monkeypatch.setattr(ops_mod, "encode_deferred", orig)
        assert (len(calls) == 2) == defer
        return y0, y.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}

    a0, a1, ga = run(True)
    b0, b1, gb = run(False)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    # gradients: to the run-to-run spread of the backward pass itself -- f32: atomics' summation order (1e-6); bf16: that
    # order flips a few bf16 roundings of the activation gradients, 5e-3 between two IDENTICAL runs
    # (tools/micro/dbg_defer_grads.py)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for n in ga:
        assert rel_l2(ga[n], gb[n]) < tol, n
    assert ga["encode_x.weight"].abs().sum() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_inference_decodes_inside_the_last_block_bit_identically(dtype, monkeypatch):
    """Without autograd the last ResnetBlock's tail kernel applies the 1x1 decoder itself (tdx_gn_apply_decode): same
    output, bit for bit, as with models.ddpm.FUSE_DECODE = False; with autograd on, the block output is kept (the decoder's weight
    gradient needs it) and the fused route is not taken."""
    from turbdiff_amd import _lib as L
    from turbdiff_amd.models import ddpm as D

    torch.manual_seed(0)
    net = D.DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                           u_net_levels=2, norm_type="group")
    x = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(1)).to(dev())
    c_local = torch.randn(4, 26, 12, 17, generator=torch.Generator().manual_seed(2)).to(dev())
    t = torch.tensor([3, 250, 499]).to(dev())
    net.to(dev()).set_compute_dtype(dtype)
    C = cond(c_local)
    seen = []
    orig = L.call
    monkeypatch.setattr(L, "call", lambda name, *a, **k: (seen.append(name), orig(name, *a, **k))[1])
    with torch.no_grad():
        fused = net(x, t, C).clone()
    assert "tdx_gn_apply_decode" in seen and "tdx_decode_fwd" not in seen
    seen.clear()
    monkeypatch.setattr(D, "FUSE_DECODE", False)
    with torch.no_grad():
        plain = net(x, t, C).clone()
    assert "tdx_decode_fwd" in seen and "tdx_gn_apply_decode" not in seen
    assert torch.equal(fused, plain)
    monkeypatch.setattr(D, "FUSE_DECODE", True)
    seen.clear()
    y = net(x, t, C)
    assert "tdx_decode_fwd" in seen and "tdx_gn_apply_decode" not in seen and y.requires_grad


OPTION_VARIANTS = {
    "instance": (dict(norm_type="instance"), {}),
    "layer": (dict(norm_type="layer"), {}),
    "gelu": (dict(actfn=torch.nn.GELU), {}),
    "l1": ({}, dict(loss_type="l1")),
    "clip": ({}, dict(clip_denoised=True)),
    "learned_var": (dict(out_features=8), dict(learned_variances=True, elbo_weight=0.001)),
    "learned_var_noelbo": (dict(out_features=8), dict(learned_variances=True)),
}


@pytest.mark.parametrize("tag", list(OPTION_VARIANTS))
def test_constructor_options_golden(golden, tag):
    """Constructor options off the shipped configuration (reference ddpm.py:399-431, 621-633) against vectors
    of the unmodified reference: eps_hat, loss, every parameter's gradient norm (small gradients in full) and,
    for clip_denoised, the sampling loop.  fp32 mode."""
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    g = golden("options")
    dm_kw, gd_kw = OPTION_VARIANTS[tag]
    dm_args = dict(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=8, u_net_levels=2,
                   norm_type="group")
    dm_args.update(dm_kw)
    gd_args = dict(timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True)
    gd_args.update(gd_kw)
    net = DenoisingModel(**dm_args)
    sd = dict(golden("model_cfg1").sub("sd/"))
    sd.update(g.sub(f"{'learned_var' if tag == 'learned_var_noelbo' else tag}/sd/"))
    net.load_state_dict(sd, strict=True)
    diff = GaussianDiffusion(net, **gd_args).to(dev())
    x, C, t = g["x"].to(dev()), cond(g["c_local"]), g["t"].to(dev())
    md = SimpleNamespace(cell_idx=g["cell_idx"].to(dev()))
    with torch.no_grad():
        eps = diff.model(x, t, C)
    assert rel_l2(eps.cpu(), g[f"{tag}/eps_hat"]) < 1e-4
    loss, _ = diff.p_losses(x, t, C, md, None, noise=g[f"{tag}/noise"].to(dev()))
    loss.backward()
    assert abs(loss.item() - g[f"{tag}/loss"].item()) < 1e-4 * abs(g[f"{tag}/loss"].item())
    for name, p in diff.model.named_parameters():
        ref = g[f"{tag}/gnorm/{name}"].item()
        got = p.grad.norm().item()
        assert abs(got - ref) < 2e-3 * ref + 2e-6, (name, got, ref)
        if f"{tag}/grad/{name}" in g.z.files:
            assert_grad_close(name, p.grad.cpu(), g[f"{tag}/grad/{name}"], 2e-3)
    if tag == "learned_var_noelbo":
        # sampling with learned variances (VERDICT r4 missing 3).  The unmodified reference raises inside its own loop
        # (recorded by the generator); pinned instead: its p_sample (mean, per-voxel lerped log_var) at t = 6 and t = 0
        # along a 10-step loop, and that loop's result with the reference's helpers around the reference's p_sample
        assert str(g.z[f"{tag}/sample_raises"]) == "RuntimeError"
        noises = [g[f"{tag}/sample_noise/{i}"].to(dev()) for i in range(int(g[f"{tag}/n_noise"]))]
        it = iter(noises)
        out = diff.p_sample_loop(x, C, md.cell_idx, noise_fn=lambda like: next(it))
        assert next(it, None) is None
        assert rel_l2(out.cpu(), g[f"{tag}/sample"]) < 1e-4
        # the per-step pieces: replay the loop up to t = 6 with the same draws
        x_t = noises[0]
        k = 1
        for t in reversed(range(10)):
            mean, log_var = diff.p_sample(x_t, t, C, md.cell_idx)
            if t in (6, 0):
                assert rel_l2(mean.cpu(), g[f"{tag}/p_sample_mean/{t}"]) < 1e-4
                assert rel_l2(log_var.cpu(), g[f"{tag}/p_sample_log_var/{t}"]) < 1e-4
            if t == 6:
                break
            x_t = mean + (log_var / 2).exp() * noises[k]
            inside = torch.zeros(x[0, 0].numel(), dtype=torch.bool, device=dev())
            inside[md.cell_idx] = True
            tt = torch.full((x.shape[0],), t, dtype=torch.long, device=dev())
            x_t = torch.where(inside.view(x.shape[-3:]), x_t, diff.q_sample(x, tt, noises[k + 1]))
            k += 2
        # the default (no injected noise) call goes through the same loop with torch.randn_like
        torch.manual_seed(5)
        assert torch.isfinite(diff.p_sample_loop(x, C, md.cell_idx)).all()
    if tag == "clip":
        it = iter([g[f"{tag}/sample_noise/{i}"].to(dev()) for i in range(int(g[f"{tag}/n_noise"]))])
        out = diff.p_sample_loop(x, C, md.cell_idx, noise_fn=lambda like: next(it))
        assert next(it, None) is None
        assert rel_l2(out.cpu(), g[f"{tag}/sample"]) < 1e-4
