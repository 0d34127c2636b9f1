"""Variance schedules and the 12 registered buffers of GaussianDiffusionModel (oracle; test infrastructure).

Follows mpd/models/diffusion_models/helpers.py:26-46 and diffusion_model_base.py:66-103 of the reference.
Checked bit-for-bit against tests/golden/schedules.npz (made by importing the reference).
"""
import numpy as np
import torch

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def _np_sqrt(x: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(np.sqrt(x.numpy()))


def exponential_betas(T: int, beta_start: float = 1e-4, beta_end: float = 1.0) -> torch.Tensor:
    # helpers.py:40-46 - note linspace(0, T, T): the last abscissa is T, spacing T/(T-1)
    x = torch.linspace(0, T, T)
    b0 = torch.tensor(beta_start, dtype=torch.float32)
    b1 = torch.tensor(beta_end, dtype=torch.float32)
    a = 1 / T * torch.log(b1 / b0)
    return b0 * torch.exp(a * x)


def cosine_betas(T: int, s: float = 0.008, a_min: float = 0.0, a_max: float = 0.999) -> torch.Tensor:
    # helpers.py:26-37 (float64 numpy, cast to fp32 at the end)
    steps = T + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.tensor(np.clip(betas, a_min, a_max), dtype=torch.float32)


def make_buffers(T: int, variance_schedule: str = "exponential") -> dict:
    """diffusion_model_base.py:66-103."""
    if variance_schedule == "exponential":
        betas = exponential_betas(T)
    elif variance_schedule == "cosine":
        betas = cosine_betas(T)
    else:
        raise NotImplementedError(variance_schedule)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(torch.clamp(post_var, min=1e-20)),
        # the reference calls np.sqrt on torch tensors here (:101,103).  numpy's fp32 sqrt is correctly rounded,
        # torch's vectorised CPU sqrt is not always (1 ulp off on alphas_cumprod_prev[67] at T=100), so the mix
        # must be reproduced for bit-exact buffers.
        "posterior_mean_coef1": betas * _np_sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * _np_sqrt(alphas) / (1.0 - ac),
    }
