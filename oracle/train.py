"""Training step (oracle; test infrastructure - never imported by the product path).

CPU restatement of what one iteration of mpd/trainer/trainer.py:186-283 computes for the GaussianDiffusionModel:
  loss, grads  = autograd of diffusion_model_base.py:331-352 (oracle/diffusion.py:p_losses over oracle/unet.py)
  clip         = torch.nn.utils.clip_grad_norm_(params, max_norm)            trainer.py:268-272
  Adam         = torch.optim.Adam(lr) defaults (betas (0.9, 0.999), eps 1e-8)  trainer.py:140
  EMA          = old * beta + (1 - beta) * new                               trainer.py:67-85
Pinned against the REAL reference (tests/golden/train.npz, written by tests/golden/make_golden.py --only train by running the
reference's own p_losses / backward / clip_grad_norm_ / Adam): tests/test_oracle_golden.py::test_training_step_vs_reference_golden.
"""
import math

import torch

from . import diffusion


def loss_and_grads(sd: dict, x_start, t, hard_conds, noise, T, variance_schedule="exponential", predict_epsilon=True, loss_type="l2",
                   dtype=torch.float32):
    """-> (loss, {name: d loss / d sd[name]}) ; sd values are leaf copies in `dtype` (float64 gives a tight reference)."""
    leaf = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    cast = lambda v: v.to(dtype) if torch.is_floating_point(v) else v
    hc = {k: cast(v) for k, v in (hard_conds or {}).items()}
    loss = diffusion.p_losses(leaf, cast(x_start), t, hc, cast(noise), T, variance_schedule, predict_epsilon, loss_type)
    names = list(leaf)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names])
    return loss.detach(), dict(zip(names, grads))


def clip_grad_norm(grads: dict, max_norm: float):
    """torch.nn.utils.clip_grad_norm_: -> (total_norm, clipped grads)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adam_step(params: dict, grads: dict, state: dict, lr: float, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam.step (no weight decay, no amsgrad); state = {'step': int, 'm': {...}, 'v': {...}} updated in place."""
    state["step"] = state.get("step", 0) + 1
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** state["step"], 1 - b2 ** state["step"]
    out = {}
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m", {}).get(k, torch.zeros_like(p)) * b1 + (1 - b1) * g
        v = state.setdefault("v", {}).get(k, torch.zeros_like(p)) * b2 + (1 - b2) * g * g
        state["m"][k], state["v"][k] = m, v
        out[k] = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return out


def ema_update(ema: dict, params: dict, beta: float):
    return {k: ema[k] * beta + (1 - beta) * params[k] for k in ema}
