"""DDPM reverse step, guidance inner loop and the planning loop (oracle; test infrastructure).

Follows mpd/models/diffusion_models/sample_functions.py:5-83 and diffusion_model_base.py:121-182,285-316.
Noise is INJECTED (``noise[0]`` = the initial x ~ N(0,I), ``noise[1+k]`` = the randn_like of the k-th loop
iteration) so that a CPU oracle run and a GPU run see the same stream (SURVEY.md section 7 "RNG").
"""
import torch

from . import schedules as _sched
from .unet import unet_forward


def apply_hard_conditioning(x: torch.Tensor, hard_conds: dict) -> torch.Tensor:
    # sample_functions.py:5-8 (in place)
    for k, v in hard_conds.items():
        x[:, k, :] = v.clone()
    return x


def predict_start_from_noise(buf: dict, x: torch.Tensor, t: int, eps: torch.Tensor, predict_epsilon: bool = True):
    # diffusion_model_base.py:121-132
    if predict_epsilon:
        return buf["sqrt_recip_alphas_cumprod"][t] * x - buf["sqrt_recipm1_alphas_cumprod"][t] * eps
    return eps


def p_mean(buf: dict, sd: dict, x: torch.Tensor, t: int, predict_epsilon: bool = True, eps_fn=None) -> torch.Tensor:
    # diffusion_model_base.py:143-155 + q_posterior :134-141
    B = x.shape[0]
    tt = torch.full((B,), t, dtype=torch.long)
    eps = eps_fn(x, tt) if eps_fn is not None else unet_forward(sd, x, tt)
    x0 = predict_start_from_noise(buf, x, t, eps, predict_epsilon)
    x0 = x0.clamp(-1.0, 1.0)
    return buf["posterior_mean_coef1"][t] * x0 + buf["posterior_mean_coef2"][t] * x


def guide_gradient_steps(x: torch.Tensor, hard_conds: dict, guide, n_guide_steps: int, scale_grad_by_std: bool = False,
                         model_var=None) -> torch.Tensor:
    # sample_functions.py:65-83 (inference.py:240-244 leaves scale_grad_by_std at its default False)
    for _ in range(n_guide_steps):
        g = guide(x)
        if scale_grad_by_std:  # :77-78
            g = model_var * g
        x = x + g
        x = apply_hard_conditioning(x, hard_conds)
    return x


def ddpm_step(buf: dict, sd: dict, x: torch.Tensor, hard_conds: dict, i: int, noise: torch.Tensor,
              guide=None, n_guide_steps: int = 1, t_start_guide: float = float("inf"),
              noise_std: float = 1.0, predict_epsilon: bool = True, eps_fn=None, scale_grad_by_std: bool = False) -> torch.Tensor:
    """One ddpm_sample_fn call at loop index i (may be negative). sample_functions.py:17-62."""
    t = max(i, 0)  # :28-30
    x = p_mean(buf, sd, x, t, predict_epsilon, eps_fn)
    std = torch.exp(0.5 * buf["posterior_log_variance_clipped"][t])  # :35-36
    if guide is not None and i < t_start_guide:  # :39 compares t_single (the un-clamped index)
        model_var = torch.exp(buf["posterior_log_variance_clipped"][t])  # :36
        x = guide_gradient_steps(x, hard_conds, guide, n_guide_steps, scale_grad_by_std, model_var)
    n = noise.clone()
    if t == 0:  # :52
        n.zero_()
    return x + std * n * noise_std  # :62


def p_sample_loop(buf: dict, sd: dict, hard_conds: dict, noise: torch.Tensor, T: int,
                  n_diffusion_steps_without_noise: int = 0, **kw) -> torch.Tensor:
    """diffusion_model_base.py:157-182 with return_chain=True.  Returns chain [T+n0+1, B, H, D]
    (already in run_inference's 'diffsteps b h d' order, :310)."""
    x = apply_hard_conditioning(noise[0].clone(), hard_conds)
    chain = [x.clone()]
    k = 1
    for i in reversed(range(-n_diffusion_steps_without_noise, T)):
        x = ddpm_step(buf, sd, x, hard_conds, i, noise[k], **kw)
        x = apply_hard_conditioning(x, hard_conds)
        chain.append(x.clone())
        k += 1
    return torch.stack(chain, dim=0)


def run_inference(sd: dict, hard_conds: dict, noise: torch.Tensor, T: int, variance_schedule: str = "exponential",
                  n_diffusion_steps_without_noise: int = 0, dtype=torch.float32, **kw) -> torch.Tensor:
    """diffusion_model_base.py:285-316: hard conds 'd -> b d', full chain [steps+1, B, H, D].
    dtype=float64 evaluates the same algorithm (same fp32 schedule buffers) in double: the rounding-free yardstick."""
    buf = {k: v.to(dtype) for k, v in _sched.make_buffers(T, variance_schedule).items()}
    B = noise.shape[1]
    hc = {k: (v[None, :].expand(B, -1).clone() if v.dim() == 1 else v.clone()) for k, v in hard_conds.items()}
    return p_sample_loop(buf, sd, hc, noise, T, n_diffusion_steps_without_noise, **kw)


def ddim_sample(sd: dict, hard_conds: dict, x_T: torch.Tensor, T: int, variance_schedule: str = "exponential",
                guide=None, n_guide_steps: int = 1, t_start_guide: float = float("inf"), predict_epsilon: bool = True,
                dtype=torch.float32) -> torch.Tensor:
    """diffusion_model_base.py:184-259 with eta = 0 (sigma = 0: the randn_like draws do not matter).  Returns the
    chain [pairs+1, B, H, D] in 'diffsteps b h d' order.  x_start is NOT clamped on this path (the reference does not)."""
    buf = {k: v.to(dtype) for k, v in _sched.make_buffers(T, variance_schedule).items()}
    B = x_T.shape[0]
    hc = {k: (v[None, :].expand(B, -1).clone() if v.dim() == 1 else v.clone()) for k, v in hard_conds.items()}
    sampling = T // 5
    times = torch.linspace(0, T - 1, steps=sampling + 1)
    times = list(reversed(torch.cat((torch.tensor([-1.0]), times)).int().tolist()))
    x = apply_hard_conditioning(x_T.clone(), hc)
    chain = [x.clone()]
    for time, time_next in zip(times[:-1], times[1:]):
        tt = torch.full((B,), time, dtype=torch.long)
        out = unet_forward(sd, x, tt)
        a_t, b_t = buf["sqrt_recip_alphas_cumprod"][time], buf["sqrt_recipm1_alphas_cumprod"][time]
        if predict_epsilon:
            x_start, pred_noise = a_t * x - b_t * out, out
        else:
            x_start, pred_noise = out, (a_t * x - out) / b_t
        if time_next < 0:
            x = apply_hard_conditioning(x_start, hc)
            chain.append(x.clone())
            break
        alpha_next = buf["alphas_cumprod"][time_next]
        c = (1 - alpha_next).sqrt()
        x = x_start * alpha_next.sqrt() + c * pred_noise
        if guide is not None and time_next < t_start_guide:
            # the reference names n_guide_steps in ddim_sample's signature but does not forward it to guide_gradient_steps
            # (diffusion_model_base.py:240-246): always ONE guide step per time pair
            x = guide_gradient_steps(x, hc, guide, 1)
        x = apply_hard_conditioning(x, hc)
        chain.append(x.clone())
    return torch.stack(chain, dim=0)


# ------------------------------------------------------------------------------------------------ forward loss
def q_sample(buf: dict, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """diffusion_model_base.py:320-330 (extract = gather + reshape to [B,1,1], sample_functions.py:11-14)."""
    a = buf["sqrt_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1)
    b = buf["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1)
    return a * x_start + b * noise


def p_losses(sd: dict, x_start: torch.Tensor, t: torch.Tensor, hard_conds: dict, noise: torch.Tensor, T: int,
             variance_schedule: str = "exponential", predict_epsilon: bool = True, loss_type: str = "l2") -> torch.Tensor:
    """diffusion_model_base.py:331-352 with the unweighted WeightedL1 / WeightedL2 of helpers.py:71-99 (loss.mean())."""
    buf = _sched.make_buffers(T, variance_schedule)
    x_noisy = apply_hard_conditioning(q_sample(buf, x_start, t, noise), hard_conds)
    x_recon = apply_hard_conditioning(unet_forward(sd, x_noisy, t), hard_conds)
    targ = noise if predict_epsilon else x_start
    err = (x_recon - targ).abs() if loss_type == "l1" else torch.nn.functional.mse_loss(x_recon, targ, reduction="none")
    return err.mean()
