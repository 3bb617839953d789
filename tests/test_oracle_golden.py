"""Pin the CPU oracle (oracle/turbdiff_oracle.py) against the golden vectors that
tests/golden/make_golden.py generated from the unmodified reference."""

import numpy as np
import pytest
import torch

from conftest import assert_grad_close, rel_l2
from oracle import turbdiff_oracle as O

TOL = 2e-6  # fp32 rel-L2 between two CPU evaluation orders of the same math


@pytest.mark.parametrize("name", ["linear", "log-linear", "log-snr-linear", "cosine", "sigmoid"])
@pytest.mark.parametrize("T", [10, 500, 1000])
def test_schedule_buffers_bit_exact(golden, name, T):
    g = golden("schedules")
    buf = O.schedule_buffers(name, T)
    assert len(buf) == 10
    for k, v in buf.items():
        ref = g[f"{name}/{T}/{k}"]
        assert v.dtype == torch.float32 and v.shape == (T,)
        # bitwise (the reference's linear schedule at T=10 has beta > 1 -> NaNs, kept as is)
        assert torch.equal(v.view(torch.int32), ref.view(torch.int32)), f"{name}/{T}/{k} differs"


def test_log_snr_schedule_is_a_sigmoid():
    # size-independent property: logit(abar_t) is linear in t between log 1e3 and log 1e-5
    T = 500
    abar = torch.cumprod(1 - O.beta_schedule("log-snr-linear", T), 0)
    level = torch.linspace(np.log(1e3), np.log(1e-5), T, dtype=torch.float64)
    assert torch.allclose(torch.logit(abar), level, atol=1e-5)


@pytest.mark.parametrize("T", [10, 500])
def test_time_features(golden, T):
    g = golden("ops")
    y = O.time_features(g[f"tfreq/{T}/t"], g[f"tfreq/{T}/y"].shape[-1], T)
    assert torch.equal(y, g[f"tfreq/{T}/y"])


def _grads(outs, ins, gy):
    return torch.autograd.grad(outs, ins, gy, allow_unused=False)


def test_block_film(golden):
    g = golden("ops")
    sd = {k: v.clone().requires_grad_() for k, v in g.sub("block/sd/").items()}
    x, sc, sh = (g[f"block/{n}"].clone().requires_grad_() for n in ("x", "scale", "shift"))
    assert rel_l2(O.conv3_replicate(x, sd["conv.weight"], sd["conv.bias"]), g["block/conv_out"]) < TOL
    y = O.block(sd, "", x, lambda C: 8, (sc, sh))
    assert rel_l2(y, g["block/y"]) < TOL
    names = list(sd)
    gr = _grads(y, [x, sc, sh] + [sd[n] for n in names], g["block/gy"])
    assert rel_l2(gr[0], g["block/gx"]) < TOL
    assert rel_l2(gr[1], g["block/gscale"]) < TOL
    assert rel_l2(gr[2], g["block/gshift"]) < TOL
    for n, gv in zip(names, gr[3:]):
        assert_grad_close(n, gv, g[f"block/grad/{n}"], 1e-5)


def test_block_nofilm(golden):
    g = golden("ops")
    sd = g.sub("block/sd/")
    x = g["block_nofilm/x"].clone().requires_grad_()
    y = O.block(sd, "", x, lambda C: 8)
    assert rel_l2(y, g["block_nofilm/y"]) < TOL
    assert rel_l2(_grads(y, [x], g["block/gy"])[0], g["block_nofilm/gx"]) < TOL


@pytest.mark.parametrize("tag", ["resnet_proj", "resnet_id"])
def test_resnet_block(golden, tag):
    g = golden("ops")
    sd = {k: v.clone().requires_grad_() for k, v in g.sub(f"{tag}/sd/").items()}
    x, c = g[f"{tag}/x"].clone().requires_grad_(), g[f"{tag}/c"].clone().requires_grad_()
    y = O.resnet_block(sd, "", x, c, lambda C: 8)
    assert rel_l2(y, g[f"{tag}/y"]) < TOL
    names = list(sd)
    gr = _grads(y, [x, c] + [sd[n] for n in names], g[f"{tag}/gy"])
    assert rel_l2(gr[0], g[f"{tag}/gx"]) < TOL
    assert rel_l2(gr[1], g[f"{tag}/gc"]) < 1e-5
    for n, gv in zip(names, gr[2:]):
        assert_grad_close(n, gv, g[f"{tag}/grad/{n}"], 1e-5)


def test_attention(golden):
    g = golden("ops")
    assert rel_l2(O.sdpa(g["sdpa/q"], g["sdpa/k"], g["sdpa/v"]), g["sdpa/o"]) < TOL
    sd = {k: v.clone().requires_grad_() for k, v in g.sub("attn/sd/").items()}
    x = g["attn/x"].clone().requires_grad_()
    xn = O.group_norm(x, 8, sd["fn.norm.weight"], sd["fn.norm.bias"])
    y = O.attention(sd, "fn.fn.", xn) + x
    assert rel_l2(y, g["attn/y"]) < TOL
    names = list(sd)
    gr = _grads(y, [x] + [sd[n] for n in names], g["attn/gy"])
    assert rel_l2(gr[0], g["attn/gx"]) < TOL
    for n, gv in zip(names, gr[1:]):
        assert_grad_close(n, gv, g[f"attn/grad/{n}"], 1e-5)


def test_resize_and_unet_wiring(golden):
    g = golden("ops")
    x = g["resize/x"]
    d = O.resize(x, O.down_size(x.shape[-3:]))
    assert list(d.shape[-3:]) == [6, 3, 3]
    assert rel_l2(d, g["resize/down"]) < TOL
    assert rel_l2(O.resize(d, x.shape[-3:]), g["resize/up"]) < TOL
    # the parameter-free UNet of the fixture: blocks = scale * first 3 channels
    for tag in ["unet_interp_a", "unet_interp_b"]:
        x = g[f"{tag}/x"].clone().requires_grad_()
        h, skips = x, []
        for s in (1.5, 0.5):
            h = s * h[:, :3]
            skips.append(h)
            h = O.resize(h, O.down_size(h.shape[-3:]))
        h = 3.0 * h[:, :3]
        for s in (2.0, -1.0):
            sk = skips.pop()
            h = s * torch.cat((O.resize(h, sk.shape[-3:]), sk), dim=1)[:, :3]
        assert rel_l2(h, g[f"{tag}/y"]) < TOL
        assert rel_l2(_grads(h, [x], g[f"{tag}/gy"])[0], g[f"{tag}/gx"]) < TOL


def test_where_cells_and_q_sample(golden):
    g = golden("ops")
    a, b, idx = g["where/a"], g["where/b"], g["where/idx"]
    assert torch.equal(O.where_cells(idx, a, b), g["where/ab"])
    assert torch.equal(O.where_cells(idx, a), g["where/a0"])
    buf = O.schedule_buffers("log-snr-linear", 10)
    assert torch.equal(O.q_sample(buf, a, g["qsample/t"], b), g["qsample/y"])


def test_denoiser_cfg1(golden):
    g = golden("model_cfg1")
    sd = g.sub("sd/")
    y = O.denoiser(sd, g["x"], g["t"], g["c_local"], timesteps=10)
    assert rel_l2(y, g["eps_hat"]) < TOL
    g2 = golden("model_cfg1_48")
    y = O.denoiser(sd, g2["x"].float(), g2["t"], g2["c_local"].float(), timesteps=10)
    assert rel_l2(y, g2["eps_hat"]) < TOL


@pytest.mark.parametrize("nb", [1, 0])
def test_p_losses_and_grads_cfg1(golden, nb):
    g = golden("model_cfg1")
    sd = {k: v.clone().requires_grad_() for k, v in g.sub("sd/").items()}
    buf = O.schedule_buffers("log-snr-linear", 10)
    loss, _ = O.p_losses(sd, buf, g["x"], g["t"], g["c_local"], g["cell_idx"], g[f"loss_nb{nb}/noise"],
                         timesteps=10, noise_bcs=bool(nb))
    assert abs(loss.item() - g[f"loss_nb{nb}/loss"].item()) < 1e-6 * abs(loss.item())
    names = list(sd)
    gr = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gv in zip(names, gr):
        assert_grad_close(n, gv, g[f"loss_nb{nb}/grad/{n}"], 2e-5)


@pytest.mark.parametrize("tag,nb,start", [("nb1", True, None), ("nb0", False, None), ("nb1_from5", True, 5)])
def test_p_sample_loop_cfg1(golden, tag, nb, start):
    g = golden("sample_cfg1")
    sd = golden("model_cfg1").sub("sd/")
    buf = O.schedule_buffers("log-snr-linear", 10)
    n = int(g[f"{tag}/n_noise"])
    noises = [g[f"{tag}/noise/{i}"] for i in range(n)]
    with torch.no_grad():
        out = O.p_sample_loop(sd, buf, g["x_bcs"], g["c_local"], g["cell_idx"], noises, timesteps=10,
                              noise_bcs=nb, start_from=start)
    assert rel_l2(out, g[f"{tag}/out"]) < 1e-5
    # draw count: 1 + per step t>0 (1 + noise_bcs)
    steps = 10 if start is None else start
    assert n == 1 + (steps - 1) * (2 if nb else 1)


def test_p_sample_mean_logvar(golden):
    g = golden("sample_cfg1")
    sd = golden("model_cfg1").sub("sd/")
    buf = O.schedule_buffers("log-snr-linear", 10)
    t = torch.full((2,), 6, dtype=torch.long)
    with torch.no_grad():
        eps = O.denoiser(sd, g["x_bcs"], t, g["c_local"], timesteps=10)
        _, mean = O.model_mean(buf, g["x_bcs"], t, eps, g["cell_idx"], True)
    assert rel_l2(mean, g["p_sample_t6/mean"]) < TOL
    assert torch.equal(buf["log_betas"][t], g["p_sample_t6/log_var"])


@pytest.mark.parametrize("tag,norm", [("instance", "instance"), ("layer", "layer")])
def test_oracle_norm_type_variants(golden, tag, norm):
    """The oracle's norm_type switch (GroupNorm with G = C / G = 1, ddpm.py:424-431) against reference vectors."""
    g = golden("options")
    sd = dict(golden("model_cfg1").sub("sd/"))
    with torch.no_grad():
        eps = O.denoiser(sd, g["x"], g["t"], g["c_local"], timesteps=10, norm_type=norm)
    assert rel_l2(eps, g[f"{tag}/eps_hat"]) < 1e-5


def test_oracle_learned_variance_sampling(golden):
    """Learned variances (ddpm.py:732-741): the oracle's per-voxel log-variance and its 10-step loop against the
    reference's p_sample chained by the generator (the reference's own loop raises -- also recorded)."""
    g = golden("options")
    tag = "learned_var_noelbo"
    assert str(g.z[f"{tag}/sample_raises"]) == "RuntimeError"
    sd = dict(golden("model_cfg1").sub("sd/"))
    sd.update(g.sub("learned_var/sd/"))
    buf = O.schedule_buffers("log-snr-linear", 10)
    noises = [g[f"{tag}/sample_noise/{i}"] for i in range(int(g[f"{tag}/n_noise"]))]
    with torch.no_grad():
        out = O.p_sample_loop_learned_var(sd, buf, g["x"], g["c_local"], g["cell_idx"], noises, timesteps=10, noise_bcs=True)
        assert rel_l2(out, g[f"{tag}/sample"]) < 1e-5
        # t = 9 is the first step: x_t = noises[0]; check the log-variance formula on a step we can reach directly
        tt = torch.full((2,), 9, dtype=torch.long)
        _, vw = O.denoiser(sd, noises[0], tt, g["c_local"], timesteps=10).chunk(2, dim=1)
        lv = O.learned_log_var(buf, tt, vw)
        assert lv.shape == noises[0].shape
        lo = torch.minimum(O._bc(buf["log_betas"][tt], lv), O._bc(buf["posterior_log_var"][tt], lv))
        hi = torch.maximum(O._bc(buf["log_betas"][tt], lv), O._bc(buf["posterior_log_var"][tt], lv))
        assert ((lv >= lo - 1e-6) & (lv <= hi + 1e-6)).all()


# --------------------------------------------------------------------------- baseline conv layers


@pytest.mark.parametrize("tag", ["dil", "dil8", "conv_s2", "conv_k5", "deconv"])
def test_baseline_conv_oracle(golden, tag):
    """oracle/baselines_oracle.py against the reference's DilatedCNNBlock / tfnet.conv / tfnet.deconv
    (tests/golden/make_golden_baselines.py): forward, input gradient and every parameter gradient."""
    from oracle import baselines_oracle as BO

    g = golden("baselines")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k)
          for k, v in g.sub(f"{tag}/sd/").items()}
    x = g[f"{tag}/x"].clone().requires_grad_()
    if tag.startswith("dil"):
        y = BO.dilated_block(sd, x, [int(d) for d in g[f"{tag}/dilations"]])
    elif tag == "conv_s2":
        y = BO.tfnet_conv(sd, x, 3, 2, training=True)
    elif tag == "conv_k5":
        y = BO.tfnet_conv(sd, x, 5, 2, training=False)
    else:
        y = BO.tfnet_deconv(sd, x)
    y.backward(g[f"{tag}/gy"])
    assert rel_l2(y, g[f"{tag}/y"]) < 1e-6 and rel_l2(x.grad, g[f"{tag}/gx"]) < 1e-5
    for k in g.keys(f"{tag}/grad/"):
        name = k[len(f"{tag}/grad/"):]
        assert rel_l2(sd[name].grad, g[k]) < 1e-5, name
