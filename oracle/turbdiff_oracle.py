"""CPU oracle for the turbdiff denoising-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

A functional restatement (stock PyTorch-CPU ops, fp32, NCDHW) of what the reference's
``turbdiff/models/ddpm.py`` computes on the hot path: the DenoisingModel 3D U-Net and the
GaussianDiffusion q_sample / p_losses / p_sample / p_sample_loop arithmetic.  It works on a
plain ``state_dict`` (the reference's key schema) instead of nn.Modules.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
the golden vectors in ``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` produced by
running the unmodified reference in the build container.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does.

Reference citations are ``file:line`` relative to the upstream repository.
"""

from __future__ import annotations

import math

import numpy as np
import scipy.optimize
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- schedules


def beta_schedule(name: str, T: int) -> torch.Tensor:
    """fp64 betas.  ddpm.py:511-594 (five schedules), dispatch ddpm.py:643-656."""
    if name == "linear":  # ddpm.py:511-518
        s = 1000 / T
        return torch.linspace(s * 1e-4, s * 0.02, T, dtype=torch.float64)
    if name == "log-linear":  # ddpm.py:521-536
        k = np.arange(1, T + 1)
        target = np.log(1e-6)

        def resid(a_T):
            return np.log(T + k * (a_T - 1)).sum() - T * np.log(T) - target

        a_T = scipy.optimize.bisect(resid, 1e-10, 1.0)
        return torch.tensor(1 - (T + k * (a_T - 1)) / T)
    if name == "log-snr-linear":  # ddpm.py:539-563
        lo, hi = np.log(1e3), np.log(1e-5)
        abar = np.empty(T)
        for i in range(T):
            tt = i + 1
            level = ((T - tt) * lo + (tt - 1) * hi) / (T - 1)

            def resid(a, level=level):
                return np.log(a) - np.log1p(-a) - level

            abar[i] = scipy.optimize.bisect(resid, 1e-8, 1.0 - 1e-8)
        alphas = np.concatenate((abar[:1], abar[1:] / abar[:-1]))
        return torch.tensor(1 - alphas)
    if name in ("cosine", "sigmoid"):
        u = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        if name == "cosine":  # ddpm.py:566-576
            s = 0.008
            abar = torch.cos((u + s) / (1 + s) * math.pi * 0.5) ** 2
        else:  # ddpm.py:579-594, start=-3, end=3, tau=1
            lo_, hi_ = torch.tensor(-3.0).sigmoid(), torch.tensor(3.0).sigmoid()
            abar = (-((u * 6 - 3)).sigmoid() + hi_) / (hi_ - lo_)
        abar = abar / abar[0]
        return torch.clip(1 - abar[1:] / abar[:-1], 0, 0.999)
    raise ValueError(f"unknown beta schedule {name}")


def schedule_buffers(name: str, T: int) -> dict[str, torch.Tensor]:
    """The ten fp32 buffers of GaussianDiffusion.__init__ (ddpm.py:656-709).

    Everything is derived in fp64 and cast once, except ``posterior_log_var`` which starts
    from the already-rounded fp32 ``log_betas`` (ddpm.py:688-692) and gets its first entry
    patched (ddpm.py:697-699).
    """
    beta = beta_schedule(name, T)
    alpha = 1.0 - beta
    abar = torch.cumprod(alpha, 0)
    abar_prev = torch.cat((torch.ones(1, dtype=abar.dtype), abar[:-1]))
    f32 = lambda v: v.to(torch.float32)
    out = {
        "betas": f32(beta),
        "alphas_cumprod": f32(abar),
        "sqrt_alphas_cumprod": f32(abar.sqrt()),
        "sqrt_one_minus_alphas_cumprod": f32((1.0 - abar).sqrt()),
        "sqrt_recip_alphas_cumprod": f32(abar.rsqrt()),
        "sqrt_recipm1_alphas_cumprod": f32((1.0 / abar - 1).sqrt()),
        "log_betas": f32(beta.log()),
    }
    plv = out["log_betas"] + torch.log1p(-abar_prev) - torch.log1p(-abar)
    plv[0] = out["log_betas"][0] * (plv[1] / out["log_betas"][1])
    out["posterior_log_var"] = f32(plv)
    out["posterior_mean_coef1"] = f32(beta * abar_prev.sqrt() / (1.0 - abar))
    out["posterior_mean_coef2"] = f32((1.0 - abar_prev) * alpha.sqrt() / (1.0 - abar))
    return out


# --------------------------------------------------------------------------- building blocks


def time_features(t: torch.Tensor, dim: int, T: int) -> torch.Tensor:
    """NyquistFrequencyEmbedding (ddpm.py:122-148): sin(bias + scale * t)."""
    k = dim // 2
    phi = (1 + np.sqrt(5)) / 2
    freq = np.geomspace(1 / 8, (T / 2) / (2 * phi), num=k)
    scale = torch.tensor(np.repeat(2 * np.pi * freq / T, 2), dtype=torch.float32, device=t.device)
    bias = torch.tensor(np.tile(np.array([0, np.pi / 2]), k), dtype=torch.float32, device=t.device)
    return torch.addcmul(bias, scale, t[..., None]).sin()


def conv3_replicate(x, w, b=None):
    """3x3x3 cross-correlation with edge-clamped halo (ddpm.py:164)."""
    return F.conv3d(F.pad(x, (1,) * 6, mode="replicate"), w, b)


def group_norm(x, groups, gamma, beta, eps=1e-5):
    """nn.GroupNorm: biased variance over (C/G, X, Y, Z) per sample (ddpm.py:424-431)."""
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1)
    mu = xg.mean(-1, keepdim=True)
    var = xg.var(-1, unbiased=False, keepdim=True)
    xh = ((xg - mu) * torch.rsqrt(var + eps)).reshape(x.shape)
    return xh * gamma.view(1, C, 1, 1, 1) + beta.view(1, C, 1, 1, 1)


def norm_groups(norm_type: str, C: int) -> int:
    return {"instance": C, "layer": 1, "group": 8}[norm_type]


def block(sd, pre, x, groups, scale_shift=None):
    """Block.forward (ddpm.py:168-177): conv -> norm -> [FiLM] -> SiLU."""
    h = conv3_replicate(x, sd[pre + "conv.weight"], sd[pre + "conv.bias"])
    h = group_norm(h, groups(h.shape[1]), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    if scale_shift is not None:
        scale, shift = scale_shift
        h = shift + (scale + 1) * h
    return F.silu(h)


def resnet_block(sd, pre, x, c, groups):
    """ResnetBlock.forward (ddpm.py:190-197)."""
    ss = F.linear(c, sd[pre + "project_onto_scale_shift.weight"], sd[pre + "project_onto_scale_shift.bias"])
    scale, shift = ss[..., None, None, None].chunk(2, dim=-4)
    h = block(sd, pre + "block1.", x, groups, (scale, shift))
    h = block(sd, pre + "block2.", h, groups)
    if pre + "conv.weight" in sd:
        x = F.conv3d(x, sd[pre + "conv.weight"], sd[pre + "conv.bias"])
    return h + x


def sdpa(q, k, v):
    """softmax(q k^T / sqrt(d)) v on (b, h, n, d)  (attention.py:9-15)."""
    s = torch.einsum("bhid,bhjd->bhij", q, k) / math.sqrt(q.shape[-1])
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


def attention(sd, pre, x, heads=4):
    """Attention.forward (ddpm.py:295-308): channel thirds, head-major channels."""
    B, C, X, Y, Z = x.shape
    qkv = F.conv3d(x, sd[pre + "to_qkv.weight"])
    hd = qkv.shape[1] // 3 // heads
    q, k, v = (p.reshape(B, heads, hd, X * Y * Z).transpose(-1, -2) for p in qkv.chunk(3, dim=1))
    o = sdpa(q, k, v).transpose(-1, -2).reshape(B, heads * hd, X, Y, Z)
    return F.conv3d(o, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def resize(x, size):
    """trilinear, align_corners=True (ddpm.py:359-361, 367-369)."""
    return F.interpolate(x, size=list(size), mode="trilinear", align_corners=True)


def down_size(shape):
    """ddpm.py:358 -- halve, but never below the kernel size of 3."""
    return [max(int(s * 0.5), 3) for s in shape]


def unet(sd, pre, x, c, levels, groups):
    """UNet.forward (ddpm.py:351-372) with the DenoisingModel's blocks (ddpm.py:462-475)."""
    skips = []
    for i in range(levels):
        x = resnet_block(sd, f"{pre}downsampling_blocks.{i}.", x, c, groups)
        skips.append(x)
        x = resize(x, down_size(x.shape[-3:]))
    x = resnet_block(sd, f"{pre}center_block.0.", x, c, groups)
    npre = f"{pre}center_block.1.fn."
    xn = group_norm(x, groups(x.shape[1]), sd[npre + "norm.weight"], sd[npre + "norm.bias"])
    x = attention(sd, npre + "fn.", xn) + x
    x = resnet_block(sd, f"{pre}center_block.2.", x, c, groups)
    for i in range(levels):
        skip = skips.pop()
        x = resize(x, skip.shape[-3:])
        x = resnet_block(sd, f"{pre}upsampling_blocks.{i}.", torch.cat((x, skip), dim=-4), c, groups)
    return x


def denoiser(sd, x, t, c_local, *, timesteps, norm_type="group", pre=""):
    """DenoisingModel.forward (ddpm.py:477-505), no global conditioning / geometry embedding.

    ``c_local`` is the unbatched (c, X, Y, Z) local conditioning (ddpm.py:481, 496-501).
    dim and u_net_levels are read off the state_dict.
    """
    dim = sd[pre + "encode_x.weight"].shape[0]
    levels = sum(1 for k in sd if k.startswith(pre + "u_net.downsampling_blocks.") and k.endswith("block1.conv.weight"))
    groups = lambda C: norm_groups(norm_type, C)
    c = time_features(t, dim, timesteps)
    c = F.silu(F.linear(c, sd[pre + "process_c.0.weight"], sd[pre + "process_c.0.bias"]))
    c = F.silu(F.linear(c, sd[pre + "process_c.2.weight"], sd[pre + "process_c.2.bias"]))
    h = F.conv3d(x, sd[pre + "encode_x.weight"], sd[pre + "encode_x.bias"])
    if c_local is not None:
        e = F.conv3d(c_local[None], sd[pre + "encode_c_local.weight"], sd[pre + "encode_c_local.bias"])
        h = torch.cat((h, e.expand(h.shape[0], -1, -1, -1, -1)), dim=-4)
    h = unet(sd, pre + "u_net.", h, c, levels, groups)
    h = resnet_block(sd, pre + "decode.0.", h, c, groups)
    return F.conv3d(h, sd[pre + "decode.1.weight"], sd[pre + "decode.1.bias"])


# --------------------------------------------------------------------------- diffusion arithmetic


def _bc(v, like):
    """broadcast_right (utils.py:8-11)."""
    return v.reshape(*v.shape, *((1,) * (like.ndim - v.ndim)))


def where_cells(cell_idx, vals, other=None):
    """utils.py:22-28: vals at cell_idx, other (or 0) elsewhere."""
    out = torch.zeros_like(vals) if other is None else other.clone()
    out.flatten(-3)[..., cell_idx] = vals.flatten(-3)[..., cell_idx]
    return out


def q_sample(buf, x0, t, noise):
    """ddpm.py:818-822."""
    return _bc(buf["sqrt_alphas_cumprod"][t], x0) * x0 + _bc(buf["sqrt_one_minus_alphas_cumprod"][t], x0) * noise


def model_mean(buf, x_t, t, eps_hat, cell_idx, noise_bcs, clip=False):
    """x0-hat and posterior mean (ddpm.py:711-728, 745-752).  Returns (x0_hat, mean)."""
    x0 = _bc(buf["sqrt_recip_alphas_cumprod"][t], x_t) * x_t - _bc(buf["sqrt_recipm1_alphas_cumprod"][t], x_t) * eps_hat
    if not noise_bcs:
        x0 = where_cells(cell_idx, x0, x_t)
    if clip:
        x0 = x0.clamp(-1.0, 1.0)
    mean = _bc(buf["posterior_mean_coef1"][t], x_t) * x0 + _bc(buf["posterior_mean_coef2"][t], x_t) * x_t
    return x0, mean


def p_losses(sd, buf, x0, t, c_local, cell_idx, noise, *, timesteps, noise_bcs, norm_type="group", loss="l2"):
    """GaussianDiffusion.p_losses (ddpm.py:833-852) with the noise injected."""
    x_t = q_sample(buf, x0, t, noise)
    if not noise_bcs:
        x_t = where_cells(cell_idx, x_t, x0)
    eps_hat = denoiser(sd, x_t, t, c_local, timesteps=timesteps, norm_type=norm_type)
    err = (eps_hat - noise) ** 2 if loss == "l2" else (eps_hat - noise).abs()
    return err.flatten(-3)[..., cell_idx].flatten(1).mean(1).mean(), eps_hat


def p_sample_loop(sd, buf, x_bcs, c_local, cell_idx, noises, *, timesteps, noise_bcs, norm_type="group",
                  start_from=None, denoise_fn=None):
    """GaussianDiffusion.p_sample_loop (ddpm.py:767-816).

    ``noises`` is an iterator yielding the tensors torch.randn_like would have drawn, in the
    reference's draw order: the initial x_T, then per step t>0: z, and if noise_bcs: z'.
    """
    noises = iter(noises)
    B = x_bcs.shape[0]
    if denoise_fn is None:
        denoise_fn = lambda x, t: denoiser(sd, x, t, c_local, timesteps=timesteps, norm_type=norm_type)
    if start_from is None:
        x_t = next(noises)
        T = timesteps
    else:
        tt = torch.full((B,), start_from - 1, dtype=torch.long, device=x_bcs.device)
        x_t = q_sample(buf, x_bcs, tt, next(noises))
        T = start_from
    if not noise_bcs:
        x_t = where_cells(cell_idx, x_t, x_bcs)
    for step in reversed(range(T)):
        tt = torch.full((B,), step, dtype=torch.long, device=x_bcs.device)
        _, mean = model_mean(buf, x_t, tt, denoise_fn(x_t, tt), cell_idx, noise_bcs)
        if step == 0:
            x_t = mean
            break
        z = next(noises)
        if not noise_bcs:
            z = where_cells(cell_idx, z)
        x_t = mean + _bc((buf["log_betas"][tt] / 2).exp(), z) * z  # fixed-large variance, ddpm.py:743,804
        if noise_bcs:
            # BC cells are re-noised at level t (not t-1), ddpm.py:807-811
            x_t = where_cells(cell_idx, x_t, q_sample(buf, x_bcs, tt, next(noises)))
    return where_cells(cell_idx, x_t, x_bcs)  # ddpm.py:814


def learned_log_var(buf, t, variance_weights):
    """ddpm.py:733-741: per-voxel log-variance, lerp(log beta_t, posterior log-variance_t; sigmoid(weights))."""
    return torch.lerp(_bc(buf["log_betas"][t], variance_weights), _bc(buf["posterior_log_var"][t], variance_weights),
                      torch.sigmoid(variance_weights))


def p_sample_loop_learned_var(sd, buf, x_bcs, c_local, cell_idx, noises, *, timesteps, noise_bcs, norm_type="group",
                              start_from=None):
    """p_sample_loop (ddpm.py:767-816) with learned_variances=True: the model emits [eps_hat | variance weights]
    (ddpm.py:732-741).  The reference's own loop stops with a RuntimeError at ddpm.py:805 (broadcast_right of the 5-D
    per-voxel std, utils.py:11); this is the loop with the std applied as it is -- pinned against the reference's
    p_sample (which does run) chained by tests/golden/make_golden.py (`learned_var_noelbo/sample`)."""
    noises = iter(noises)
    B = x_bcs.shape[0]
    times = lambda s: torch.full((B,), s, dtype=torch.long, device=x_bcs.device)
    if start_from is None:
        x_t, T = next(noises), timesteps
    else:
        x_t, T = q_sample(buf, x_bcs, times(start_from - 1), next(noises)), start_from
    if not noise_bcs:
        x_t = where_cells(cell_idx, x_t, x_bcs)
    for step in reversed(range(T)):
        tt = times(step)
        eps_hat, vw = denoiser(sd, x_t, tt, c_local, timesteps=timesteps, norm_type=norm_type).chunk(2, dim=1)
        _, mean = model_mean(buf, x_t, tt, eps_hat, cell_idx, noise_bcs)
        if step == 0:
            x_t = mean
            break
        z = next(noises)
        if not noise_bcs:
            z = where_cells(cell_idx, z)
        x_t = mean + (learned_log_var(buf, tt, vw) / 2).exp() * z
        if noise_bcs:
            x_t = where_cells(cell_idx, x_t, q_sample(buf, x_bcs, tt, next(noises)))
    return where_cells(cell_idx, x_t, x_bcs)
