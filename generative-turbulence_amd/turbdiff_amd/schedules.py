"""Noise schedules and the derived per-timestep tables of the DDPM.

Host-side, run once at construction (reference: ddpm.py:511-594 for the five beta schedules,
ddpm.py:656-709 for the derived tables).  The tables are NOT in checkpoints (non-persistent
buffers), so they have to be reproduced bit for bit: everything is evaluated in float64 in the
same operation order as the reference and rounded to float32 once; the log-SNR schedule's
roots come from the same ``scipy.optimize.bisect`` calls the reference makes.
"""

from __future__ import annotations

import math

import numpy as np
import scipy.optimize as so
import torch

SCHEDULES = ("linear", "log-linear", "log-snr-linear", "cosine", "sigmoid")

# order of the 7 tables in the packed [7, T] array consumed by tdx_p_sample_step
PACKED_ORDER = (
    "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod",
    "posterior_mean_coef1",
    "posterior_mean_coef2",
    "log_betas",
    "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod",
)


def _from_alphas_cumprod(abar: torch.Tensor) -> torch.Tensor:
    abar = abar / abar[0]
    return torch.clip(1 - abar[1:] / abar[:-1], 0, 0.999)


def betas_for(name: str, timesteps: int) -> torch.Tensor:
    """float64 beta_t, t = 1..T."""
    T = timesteps
    if name == "linear":
        k = 1000 / T
        return torch.linspace(k * 0.0001, k * 0.02, T, dtype=torch.float64)
    if name == "log-linear":
        steps = np.arange(1, T + 1)
        goal = np.log(1e-6)
        root = so.bisect(lambda a: np.log(T + steps * (a - 1)).sum() - T * np.log(T) - goal, 1e-10, 1.0)
        return torch.tensor(1 - (T + steps * (root - 1)) / T)
    if name == "log-snr-linear":
        top, bottom = np.log(1e3), np.log(1e-5)
        abar = np.array(
            [
                so.bisect(
                    lambda a, lvl=((T - t) * top + (t - 1) * bottom) / (T - 1): np.log(a) - np.log1p(-a) - lvl,
                    1e-8,
                    1.0 - 1e-8,
                )
                for t in range(1, T + 1)
            ]
        )
        ratio = np.concatenate((abar[:1], abar[1:] / abar[:-1]))
        return torch.tensor(1 - ratio)
    if name == "cosine":
        u = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        return _from_alphas_cumprod(torch.cos((u + 0.008) / 1.008 * math.pi * 0.5) ** 2)
    if name == "sigmoid":
        u = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        start, end, tau = -3, 3, 1
        v0, v1 = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
        return _from_alphas_cumprod((-((u * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0))
    raise ValueError(f"unknown beta schedule {name}")


def diffusion_tables(name: str, timesteps: int) -> dict[str, torch.Tensor]:
    """The ten float32 [T] tables, keyed by the reference's buffer names."""
    betas = betas_for(name, timesteps)
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = torch.nn.functional.pad(abar[:-1], (1, 0), value=1.0)
    r = lambda v: v.to(torch.float32)
    tab = {
        "betas": r(betas),
        "alphas_cumprod": r(abar),
        "sqrt_alphas_cumprod": r(torch.sqrt(abar)),
        "sqrt_one_minus_alphas_cumprod": r(torch.sqrt(1.0 - abar)),
        "sqrt_recip_alphas_cumprod": r(torch.rsqrt(abar)),
        "sqrt_recipm1_alphas_cumprod": r(torch.sqrt(1.0 / abar - 1)),
        "log_betas": r(torch.log(betas)),
    }
    # float32 log_betas promoted back to float64 -- the reference's mixed-precision sum
    plv = tab["log_betas"] + torch.log1p(-abar_prev) - torch.log1p(-abar)
    plv[0] = tab["log_betas"][0] * (plv[1] / tab["log_betas"][1])
    tab["posterior_log_var"] = r(plv)
    tab["posterior_mean_coef1"] = r(betas * torch.sqrt(abar_prev) / (1.0 - abar))
    tab["posterior_mean_coef2"] = r((1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar))
    return tab


def pack_step_tables(tab: dict[str, torch.Tensor]) -> torch.Tensor:
    """[7, T] float32 array in the order tdx_p_sample_step expects."""
    return torch.stack([tab[k] for k in PACKED_ORDER]).contiguous()


def nyquist_embedding_tables(dim: int, timesteps: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(scale, bias) of the time embedding sin(bias + scale t)  (ddpm.py:122-148)."""
    assert dim % 2 == 0
    k = dim // 2
    golden = (1 + np.sqrt(5)) / 2
    freqs = np.geomspace(1 / 8, (timesteps / 2) / (2 * golden), num=k)
    scale = np.repeat(2 * np.pi * freqs / timesteps, 2)
    bias = np.tile(np.array([0, np.pi / 2]), k)
    return torch.tensor(scale, dtype=torch.float32), torch.tensor(bias, dtype=torch.float32)
