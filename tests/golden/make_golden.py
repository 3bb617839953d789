#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *reference*.

This script only runs in the build container, where the upstream reference is mounted
read-only at /root/reference.  It imports the reference's unmodified
``turbdiff.models.ddpm`` (with empty stand-in modules for third-party packages that the
reference pulls in for names only and that are not installed here: h5py, lightning,
wandb, omegaconf, ...), feeds it seeded inputs and stores inputs + outputs as ``.npz``
data files.  Nothing of the reference's source is stored; the fixtures are numbers only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.txt

The fixtures pin:
  schedules.npz      the 10 non-persistent schedule buffers (ddpm.py:656-709) for the five
                     beta schedules at T in {10, 500, 1000}
  ops.npz            per-module vectors on small odd shapes: Block (conv3 replicate + GN +
                     FiLM + SiLU), ResnetBlock (identity and 1x1 residual), Attention,
                     UNet (trilinear down/up + concat), NyquistFrequencyEmbedding,
                     where_cells, q_sample -- forward outputs and all gradients
  model_cfg1.npz     DenoisingModel(dim=8, 2 levels, T=10, GroupNorm(8)): state_dict, inputs,
                     eps_hat, p_losses loss and every parameter gradient
  sample_cfg1.npz    10-step p_sample_loop with every injected noise tensor, for
                     noise_bcs in {True, False} and a start_from=5 variant
  train_cfg1.npz     3 optimiser steps (clip 0.1 -> RAdam 1e-4 -> exp LambdaLR), losses and
                     final parameters
  options.npz        constructor options off the shipped path: norm_type instance / layer, GELU, l1 loss,
                     clip_denoised, learned variances with and without the ELBO term
  state_dict_manifest.txt   key / shape list of DiffusionTraining's state_dict
"""

import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, mod)
    return mod


def install_stubs():
    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, name):
            return _Anything()

    _stub("h5py", File=_Anything, Group=_Anything)
    pl = _stub(
        "pytorch_lightning",
        LightningModule=torch.nn.Module,
        LightningDataModule=object,
        Callback=object,
        Trainer=_Anything,
    )
    _stub("pytorch_lightning.callbacks", ModelCheckpoint=object, Callback=object)
    _stub(
        "pytorch_lightning.utilities",
        rank_zero_only=lambda f: f,
        move_data_to_device=lambda x, d: x,
    )
    _stub("pytorch_lightning.loggers", Logger=object)
    _stub("cachetools", cachedmethod=lambda *a, **k: (lambda f: f))
    _stub("lightning_utilities")
    _stub("lightning_utilities.core")
    _stub("lightning_utilities.core.apply_func", apply_to_collection=lambda *a, **k: a[0])
    _stub("more_itertools", chunked=lambda it, n: it)
    _stub("wandb", run=None)
    _stub("omegaconf", DictConfig=dict, OmegaConf=_Anything)
    _stub("ot", emd2=None)

    class _Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, name, default, **k):
            setattr(self, name, default)

    _stub("torchmetrics", Metric=_Metric)
    _stub("deadpool", Deadpool=_Anything)
    del pl


def to_np(t):
    return t.detach().cpu().numpy()


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(True)

    from turbdiff.models import ddpm as R
    from turbdiff.models.conditioning import Conditioning
    from turbdiff.models.utils import where_cells
    from torch import nn

    CT = Conditioning.Type.CELL_TYPE

    # ------------------------------------------------------------------ schedules
    sched = {}
    buf_names = [
        "betas",
        "alphas_cumprod",
        "sqrt_alphas_cumprod",
        "sqrt_one_minus_alphas_cumprod",
        "sqrt_recip_alphas_cumprod",
        "sqrt_recipm1_alphas_cumprod",
        "log_betas",
        "posterior_log_var",
        "posterior_mean_coef1",
        "posterior_mean_coef2",
    ]
    for name in ["linear", "log-linear", "log-snr-linear", "cosine", "sigmoid"]:
        for T in [10, 500, 1000]:
            gd = R.GaussianDiffusion(nn.Identity(), timesteps=T, beta_schedule=name)
            for b in buf_names:
                sched[f"{name}/{T}/{b}"] = to_np(getattr(gd, b))
    np.savez_compressed(OUT / "schedules.npz", **sched)

    # ------------------------------------------------------------------ per-module vectors
    ops = {}

    def gn8(c):
        return nn.GroupNorm(8, c)

    def save_mod(prefix, mod):
        for k, v in mod.state_dict().items():
            ops[f"{prefix}/sd/{k}"] = to_np(v)

    def save_grads(prefix, mod):
        for k, p in mod.named_parameters():
            ops[f"{prefix}/grad/{k}"] = to_np(p.grad)

    g = torch.Generator().manual_seed(20240101)

    def randn(*shape):
        return torch.randn(*shape, generator=g)

    # Block: conv3 replicate -> GN(8) -> FiLM -> SiLU  (ddpm.py:154-177)
    torch.manual_seed(1)
    blk = R.Block(8, 16, nn.SiLU, norm_klass=gn8)
    with torch.no_grad():
        blk.norm.weight.copy_(1 + 0.3 * randn(16))
        blk.norm.bias.copy_(0.2 * randn(16))
    x = randn(2, 8, 7, 5, 6).requires_grad_()
    scale = (0.5 * randn(2, 16, 1, 1, 1)).requires_grad_()
    shift = (0.5 * randn(2, 16, 1, 1, 1)).requires_grad_()
    gy = randn(2, 16, 7, 5, 6)
    y = blk(x, scale_shift=(scale, shift))
    y.backward(gy)
    save_mod("block", blk)
    save_grads("block", blk)
    ops.update(
        {
            "block/x": to_np(x),
            "block/scale": to_np(scale),
            "block/shift": to_np(shift),
            "block/gy": to_np(gy),
            "block/y": to_np(y),
            "block/gx": to_np(x.grad),
            "block/gscale": to_np(scale.grad),
            "block/gshift": to_np(shift.grad),
            # intermediate: the bare replicate-padded conv (ddpm.py:164)
            "block/conv_out": to_np(blk.conv(x)),
        }
    )
    # Block without FiLM (block2 of a ResnetBlock)
    blk.zero_grad()
    x2 = randn(2, 8, 7, 5, 6).requires_grad_()
    y2 = blk(x2)
    y2.backward(gy)
    save_grads("block_nofilm", blk)
    ops.update({"block_nofilm/x": to_np(x2), "block_nofilm/y": to_np(y2), "block_nofilm/gx": to_np(x2.grad)})

    # ResnetBlock (ddpm.py:180-197): 1x1 residual (8 -> 16) and identity residual (16 -> 16)
    for tag, cin, cout in [("resnet_proj", 8, 16), ("resnet_id", 16, 16)]:
        torch.manual_seed(2)
        rb = R.ResnetBlock(cin, cout, c_dim=8, actfn=nn.SiLU, norm_klass=gn8)
        x = randn(2, cin, 6, 5, 7).requires_grad_()
        c = randn(2, 8).requires_grad_()
        gy = randn(2, cout, 6, 5, 7)
        y = rb(x, c)
        y.backward(gy)
        save_mod(tag, rb)
        save_grads(tag, rb)
        ops.update(
            {f"{tag}/x": to_np(x), f"{tag}/c": to_np(c), f"{tag}/gy": to_np(gy),
             f"{tag}/y": to_np(y), f"{tag}/gx": to_np(x.grad), f"{tag}/gc": to_np(c.grad)}
        )

    # Attention (ddpm.py:286-308) inside Residual(PreNorm(GN, .)) (ddpm.py:472)
    torch.manual_seed(3)
    att = R.Residual(R.PreNorm(gn8(16), R.Attention(16)))
    x = randn(2, 16, 4, 3, 5).requires_grad_()
    gy = randn(2, 16, 4, 3, 5)
    y = att(x)
    y.backward(gy)
    save_mod("attn", att)
    save_grads("attn", att)
    ops.update({"attn/x": to_np(x), "attn/gy": to_np(gy), "attn/y": to_np(y), "attn/gx": to_np(x.grad)})
    # bare fused_attention on (b, h, n, d)
    q, k, v = randn(2, 4, 37, 32), randn(2, 4, 37, 32), randn(2, 4, 37, 32)
    from turbdiff.models.attention import fused_attention
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ops.update({"sdpa/q": to_np(q), "sdpa/k": to_np(k), "sdpa/v": to_np(v),
                    "sdpa/o": to_np(fused_attention(q, k, v))})

    # UNet skeleton with parameter-free blocks: isolates trilinear down/up + concat order
    class _Scale(nn.Module):
        def __init__(self, s, keep):
            super().__init__()
            self.s, self.keep = s, keep

        def forward(self, x):
            return self.s * x[:, : self.keep]

    un = R.UNet([_Scale(1.5, 3), _Scale(0.5, 3)], [_Scale(2.0, 3), _Scale(-1.0, 3)], _Scale(3.0, 3))
    for tag, shp in [("unet_interp_a", (2, 3, 13, 7, 6)), ("unet_interp_b", (1, 3, 12, 8, 9))]:
        x = randn(*shp).requires_grad_()
        y = un(x)
        gy = randn(*y.shape)
        y.backward(gy)
        ops.update({f"{tag}/x": to_np(x), f"{tag}/y": to_np(y), f"{tag}/gy": to_np(gy), f"{tag}/gx": to_np(x.grad)})
    # plain resize pairs as the UNet performs them (ddpm.py:358-369)
    x = randn(2, 5, 13, 7, 6)
    down = torch.nn.functional.interpolate(x, size=[max(int(s * 0.5), 3) for s in x.shape[-3:]],
                                           mode="trilinear", align_corners=True)
    up = torch.nn.functional.interpolate(down, size=x.shape[-3:], mode="trilinear", align_corners=True)
    ops.update({"resize/x": to_np(x), "resize/down": to_np(down), "resize/up": to_np(up)})

    # time embedding (ddpm.py:103-148)
    for T in [10, 500]:
        emb = R.NyquistFrequencyEmbedding(8 if T == 10 else 32, T)
        t = torch.arange(0, T, max(T // 10, 1))
        ops.update({f"tfreq/{T}/scale": to_np(emb.scale), f"tfreq/{T}/bias": to_np(emb.bias),
                    f"tfreq/{T}/t": to_np(t), f"tfreq/{T}/y": to_np(emb(t))})

    # where_cells / q_sample (utils.py:22-28, ddpm.py:818-822)
    a, b = randn(2, 4, 5, 4, 3), randn(2, 4, 5, 4, 3)
    idx = torch.tensor(sorted(np.random.default_rng(0).choice(60, 23, replace=False)))
    ops.update({"where/a": to_np(a), "where/b": to_np(b), "where/idx": to_np(idx),
                "where/ab": to_np(where_cells(idx, a, b)), "where/a0": to_np(where_cells(idx, a))})
    gd = R.GaussianDiffusion(nn.Identity(), timesteps=10, beta_schedule="log-snr-linear")
    tt = torch.tensor([3, 9])
    ops.update({"qsample/t": to_np(tt), "qsample/y": to_np(gd.q_sample(a, tt, b))})
    # LocalAttention (ddpm.py:232-283): windows of 2^3 on a grid that needs padding
    torch.manual_seed(4)
    la = R.LocalAttention(16, window_size=2, heads=4, dim_head=32)
    xl = randn(1, 16, 5, 4, 6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yl = la(xl)
    save_mod("local_attn", la)
    ops.update({"local_attn/x": to_np(xl), "local_attn/y": to_np(yl)})
    np.savez_compressed(OUT / "ops.npz", **ops)

    # ------------------------------------------------------------------ cfg1 model
    def make_model(noise_bcs=True, T=10, seed=0):
        torch.manual_seed(seed)
        dm = R.DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0,
                              timesteps=T, dim=8, u_net_levels=2, norm_type="group")
        # default init leaves GroupNorm affine at (1, 0); perturb so gamma/beta are exercised
        gg = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for n, p in dm.named_parameters():
                if ".norm." in n or n.endswith("fn.norm.weight") or n.endswith("fn.norm.bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=gg))
        return R.GaussianDiffusion(dm, timesteps=T, beta_schedule="log-snr-linear", loss_type="l2",
                                   noise_bcs=noise_bcs)

    def cell_index(W, H, D, obstacle):
        m = torch.zeros(W, H, D, dtype=torch.bool)
        m[1:-1, 1:-1, 1:-1] = True
        (x0, x1), (y0, y1), (z0, z1) = obstacle
        m[x0:x1, y0:y1, z0:z1] = False
        return torch.nonzero(m.flatten()).squeeze(-1)

    W, H, D = 24, 16, 16
    B = 2
    gx = torch.Generator().manual_seed(1234)
    gc = torch.Generator().manual_seed(1235)
    x0 = torch.randn(B, 4, W, H, D, generator=gx)
    C = {CT: torch.randn(4, W, H, D, generator=gc)}
    cidx = cell_index(W, H, D, ((5, 9), (4, 10), (0, 12)))
    t = torch.tensor([7, 2])

    model = make_model(noise_bcs=True)
    fx = {"x": to_np(x0), "c_local": to_np(C[CT]), "cell_idx": to_np(cidx), "t": to_np(t)}
    for k, v in model.model.state_dict().items():
        fx[f"sd/{k}"] = to_np(v)
    with torch.no_grad():
        fx["eps_hat"] = to_np(model.model(x0, t, C))

    noises = []
    orig_randn_like = torch.randn_like
    gn_ = torch.Generator().manual_seed(4321)

    def rec_randn_like(x, **kw):
        n = torch.randn(x.shape, generator=gn_, dtype=x.dtype)
        noises.append(n)
        return n

    for nb in [True, False]:
        model = make_model(noise_bcs=nb)
        noises.clear()
        torch.randn_like = rec_randn_like
        try:
            loss, _ = model.p_losses(x0, t, C, SimpleNamespace(cell_idx=cidx), None)
        finally:
            torch.randn_like = orig_randn_like
        loss.backward()
        tag = f"loss_nb{int(nb)}"
        fx[f"{tag}/noise"] = to_np(noises[0])
        fx[f"{tag}/loss"] = to_np(loss)
        for k, p in model.model.named_parameters():
            fx[f"{tag}/grad/{k}"] = to_np(p.grad)
    np.savez_compressed(OUT / "model_cfg1.npz", **fx)

    # one larger forward on the cfg1 grid 48x32x32 (B=1) -- only x, eps_hat (weights as above)
    W2, H2, D2 = 48, 32, 32
    xb = torch.randn(1, 4, W2, H2, D2, generator=gx)
    Cb = {CT: torch.randn(4, W2, H2, D2, generator=gc)}
    model = make_model(noise_bcs=True)
    with torch.no_grad():
        eb = model.model(xb, torch.tensor([4]), Cb)
    np.savez_compressed(OUT / "model_cfg1_48.npz", x=to_np(xb).astype(np.float16).astype(np.float32),
                        c_local=to_np(Cb[CT]).astype(np.float16).astype(np.float32), t=np.array([4]),
                        # inputs are stored fp16-exact to halve the file; recompute on them
                        )
    xb = torch.from_numpy(np.load(OUT / "model_cfg1_48.npz")["x"])
    Cb = {CT: torch.from_numpy(np.load(OUT / "model_cfg1_48.npz")["c_local"])}
    with torch.no_grad():
        eb = model.model(xb, torch.tensor([4]), Cb)
    np.savez_compressed(OUT / "model_cfg1_48.npz", x=to_np(xb).astype(np.float16), c_local=to_np(Cb[CT]).astype(np.float16),
                        t=np.array([4]), eps_hat=to_np(eb))

    # ------------------------------------------------------------------ sampling loop
    Ws, Hs, Ds = 12, 10, 9
    gs = torch.Generator().manual_seed(99)
    xs = torch.randn(2, 4, Ws, Hs, Ds, generator=gs)
    Cs = {CT: torch.randn(4, Ws, Hs, Ds, generator=gs)}
    cs = cell_index(Ws, Hs, Ds, ((3, 6), (2, 5), (0, 4)))
    sx = {"x_bcs": to_np(xs), "c_local": to_np(Cs[CT]), "cell_idx": to_np(cs)}
    for tag, nb, start in [("nb1", True, None), ("nb0", False, None), ("nb1_from5", True, 5)]:
        model = make_model(noise_bcs=nb)
        noises.clear()
        torch.randn_like = rec_randn_like
        try:
            out = model.p_sample_loop(xs, Cs, cs, pbar=False, start_from=start)
        finally:
            torch.randn_like = orig_randn_like
        sx[f"{tag}/out"] = to_np(out)
        sx[f"{tag}/n_noise"] = np.array(len(noises))
        for i, n in enumerate(noises):
            sx[f"{tag}/noise/{i}"] = to_np(n)
    # a single p_sample call (mean, log_var) at t=6
    model = make_model(noise_bcs=True)
    mean, log_var = model.p_sample(xs, 6, Cs, cs)
    sx["p_sample_t6/mean"] = to_np(mean)
    sx["p_sample_t6/log_var"] = to_np(log_var)
    np.savez_compressed(OUT / "sample_cfg1.npz", **sx)

    # ------------------------------------------------------------------ 3 training steps
    import math

    model = make_model(noise_bcs=True)
    opt = torch.optim.RAdam(model.parameters(), lr=1e-4)
    max_steps, lr, min_lr = 20, 1e-4, 1e-6
    schd = torch.optim.lr_scheduler.LambdaLR(
        opt, lambda step: math.exp(math.log(min_lr / lr) / max_steps * min(step, max_steps)))
    tx = {"x": to_np(xs), "c_local": to_np(Cs[CT]), "cell_idx": to_np(cs)}
    gt = torch.Generator().manual_seed(5)
    for step in range(3):
        tstep = torch.randint(0, 10, (2,), generator=gt)
        noises.clear()
        torch.randn_like = rec_randn_like
        try:
            loss, _ = model.p_losses(xs, tstep, Cs, SimpleNamespace(cell_idx=cs), None)
        finally:
            torch.randn_like = orig_randn_like
        opt.zero_grad()
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        schd.step()
        tx[f"step{step}/t"] = to_np(tstep)
        tx[f"step{step}/noise"] = to_np(noises[0])
        tx[f"step{step}/loss"] = to_np(loss)
        tx[f"step{step}/grad_norm"] = to_np(gnorm)
        tx[f"step{step}/lr_after"] = np.array(schd.get_last_lr()[0])
    for k, v in model.model.state_dict().items():
        tx[f"final_sd/{k}"] = to_np(v)
    np.savez_compressed(OUT / "train_cfg1.npz", **tx)

    # ------------------------------------------------------------------ non-default options
    # Variants of the constructor options the shipped configuration does not use (ddpm.py:399-431,
    # 621-633): other norm types, another activation, l1 loss, clipping, learned variances + ELBO term.
    # Same architecture seed as cfg1, sampling-size grid; outputs: eps_hat, loss, per-parameter gradient
    # norms (+ the small gradients in full), and the sampling result where the option affects sampling.
    def make_variant(dm_kw=None, gd_kw=None):
        dm_args = dict(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=8,
                       u_net_levels=2, norm_type="group")
        dm_args.update(dm_kw or {})
        gd_args = dict(timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True)
        gd_args.update(gd_kw or {})
        torch.manual_seed(0)
        dm = R.DenoisingModel(**dm_args)
        gg = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for n, p in dm.named_parameters():
                if ".norm." in n or n.endswith("fn.norm.weight") or n.endswith("fn.norm.bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=gg))
        return R.GaussianDiffusion(dm, **gd_args)

    ox = {"x": to_np(xs), "c_local": to_np(Cs[CT]), "cell_idx": to_np(cs)}
    tv = torch.tensor([6, 0])  # t = 0 exercises the ELBO's log-likelihood branch
    ox["t"] = to_np(tv)
    variants = {
        "instance": (dict(norm_type="instance"), {}),
        "layer": (dict(norm_type="layer"), {}),
        "gelu": (dict(actfn=nn.GELU), {}),
        "l1": ({}, dict(loss_type="l1")),
        "clip": ({}, dict(clip_denoised=True)),
        "learned_var": (dict(out_features=8), dict(learned_variances=True, elbo_weight=0.001)),
        "learned_var_noelbo": (dict(out_features=8), dict(learned_variances=True)),
    }
    base_sd = make_model(noise_bcs=True).model.state_dict()
    for tag, (dm_kw, gd_kw) in variants.items():
        model = make_variant(dm_kw, gd_kw)
        # the weights equal model_cfg1.npz's `sd/` except where the option changes a shape: store those only
        if tag != "learned_var_noelbo":  # (that one has learned_var's weights)
            for k, v in model.model.state_dict().items():
                if k not in base_sd or base_sd[k].shape != v.shape or not torch.equal(base_sd[k], v):
                    ox[f"{tag}/sd/{k}"] = to_np(v).astype(np.float32)
        with torch.no_grad():
            ox[f"{tag}/eps_hat"] = to_np(model.model(xs, tv, Cs))
        noises.clear()
        torch.randn_like = rec_randn_like
        try:
            loss, _ = model.p_losses(xs, tv, Cs, SimpleNamespace(cell_idx=cs), None)
        finally:
            torch.randn_like = orig_randn_like
        loss.backward()
        ox[f"{tag}/noise"] = to_np(noises[0])
        ox[f"{tag}/loss"] = to_np(loss)
        for k, p in model.model.named_parameters():
            ox[f"{tag}/gnorm/{k}"] = to_np(p.grad.norm())
            if p.numel() <= 512:
                ox[f"{tag}/grad/{k}"] = to_np(p.grad)
        if tag == "clip":  # (the reference's own p_sample_loop raises with learned variances: broadcast_right, utils.py:11)
            noises.clear()
            torch.randn_like = rec_randn_like
            try:
                out = model.p_sample_loop(xs, Cs, cs, pbar=False)
            finally:
                torch.randn_like = orig_randn_like
            ox[f"{tag}/sample"] = to_np(out)
            ox[f"{tag}/n_noise"] = np.array(len(noises))
            for i, n in enumerate(noises):
                ox[f"{tag}/sample_noise/{i}"] = to_np(n)
        if tag == "learned_var_noelbo":
            # Sampling with learned variances.  The reference's OWN loop cannot finish: ddpm.py:805 calls
            # broadcast_right(std, noise) with a 5-D per-voxel std and utils.py:11 then asks for reshape(-1, -1, -1, -1, -1)
            # (RuntimeError).  Recorded here as a fact; what CAN be pinned is everything model-dependent: the reference's
            # unmodified p_sample (mean, lerped per-voxel log_var; ddpm.py:732-741,758-765) at every step of a 10-step
            # loop whose remaining arithmetic (x = mean + exp(log_var / 2) z; BC re-noising with the reference's q_sample;
            # final where_cells; ddpm.py:796-814) is spelled out below with the reference's helpers.
            try:
                model.p_sample_loop(xs, Cs, cs, pbar=False)
                ox[f"{tag}/sample_raises"] = np.array("")
            except Exception as e:
                ox[f"{tag}/sample_raises"] = np.array(type(e).__name__)
            gl = torch.Generator().manual_seed(4242)
            drawn = []

            def draw(like):
                drawn.append(torch.randn(like.shape, generator=gl))
                return drawn[-1]

            with torch.no_grad():
                x_t = draw(xs)
                for step in reversed(range(10)):
                    mean, log_var = model.p_sample(x_t, step, Cs, cs)
                    if step in (6, 0):
                        ox[f"{tag}/p_sample_mean/{step}"] = to_np(mean)
                        ox[f"{tag}/p_sample_log_var/{step}"] = to_np(log_var)
                    if step == 0:
                        x_t = mean
                        break
                    x_t = mean + (log_var / 2).exp() * draw(x_t)
                    tt = torch.full((xs.shape[0],), step, dtype=torch.long)
                    x_t = where_cells(cs, x_t, model.q_sample(xs, tt, draw(xs)))
                x_t = where_cells(cs, x_t, xs)
            ox[f"{tag}/sample"] = to_np(x_t)
            ox[f"{tag}/n_noise"] = np.array(len(drawn))
            for i, n in enumerate(drawn):
                ox[f"{tag}/sample_noise/{i}"] = to_np(n)
    np.savez_compressed(OUT / "options.npz", **ox)

    # ------------------------------------------------------------------ state_dict manifest
    from turbdiff.models.diffusion import DiffusionTraining
    from turbdiff.data.ofles import Variable as V
    import turbdiff.models.diffusion as Dm

    class _NoMetrics(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    try:
        task = DiffusionTraining(Path("/tmp/none"), Path("/tmp/none"), dim=32, variables=(V.U, V.P),
                                 beta_schedule="log-snr-linear", timesteps=500, loss="l2", noise_bcs=True,
                                 optimizer="radam", norm_type="group", with_geometry_embedding=False)
        lines = [f"{k}\t{tuple(v.shape)}" for k, v in task.state_dict().items()]
        note = "# full DiffusionTraining state_dict (reference, dim=32, 4 levels)"
    except Exception as e:  # metrics need data files; fall back to the model part
        Dm.SampleMetricsCollection = _NoMetrics
        Dm.SampleStore = lambda *a, **k: None
        task = DiffusionTraining(Path("/tmp/none"), Path("/tmp/none"), dim=32, variables=(V.U, V.P),
                                 beta_schedule="log-snr-linear", timesteps=500, loss="l2", noise_bcs=True,
                                 optimizer="radam", norm_type="group", with_geometry_embedding=False)
        lines = [f"{k}\t{tuple(v.shape)}" for k, v in task.state_dict().items()]
        note = (f"# DiffusionTraining state_dict without the 8 metric buffers "
                f"(metrics not constructible offline: {type(e).__name__})")
    n_params = sum(p.numel() for p in task.model.model.parameters())
    (OUT / "state_dict_manifest.txt").write_text(
        note + f"\n# DenoisingModel parameters: {n_params}\n" + "\n".join(lines) + "\n")
    print("wrote fixtures to", OUT)
    for f in sorted(OUT.glob("*.npz")):
        print(f"  {f.name}: {f.stat().st_size / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
