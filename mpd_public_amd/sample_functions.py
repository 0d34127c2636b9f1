"""Drop-ins for mpd.models.diffusion_models.sample_functions (sample_functions.py:5-83).

Same names, arguments and return values; the arithmetic runs in libmpdx.so:
``ddpm_sample_fn`` = U-Net forward + posterior step (one mpdx_ddpm_step call), optional guide steps, noise.
"""
from __future__ import annotations

import math

import torch

from . import _lib


def apply_hard_conditioning(x, conditions):
    """x[:, t, :] = val for every (t, val) (in place, as sample_functions.py:5-8): horizon indices 0 / H-1 through the step kernels' own epilogue
    form (mpdx_add_noise), any other set of indices through the scatter kernel mpdx_hard_conds - python indexing semantics (negative indices,
    a later entry wins on a repeated index), values [D] or [B, D]."""
    if not conditions:
        return x
    if not x.is_cuda:
        raise RuntimeError("apply_hard_conditioning runs on the GPU (libmpdx); there is no CPU fallback")
    if not (x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 3):   # a strided view / another dtype: indexed device writes (plumbing)
        for k, v in conditions.items():
            x[:, k, :] = torch.as_tensor(v).to(device=x.device, dtype=x.dtype)
        return x
    B, H, D = x.shape
    vals = {}
    for k, v in conditions.items():
        v = torch.as_tensor(v).to(device=x.device, dtype=torch.float32)
        if v.dim() == 1:
            v = v.reshape(1, -1).expand(B, -1)
        if tuple(v.shape) != (B, D):
            raise ValueError(f"hard condition at horizon index {k}: expected [{D}] or [{B},{D}], got {tuple(v.shape)}")
        vals[int(k)] = v.contiguous()
    if set(vals) <= {0, H - 1}:
        _lib.check(_lib.load().mpdx_add_noise(x.data_ptr(), None, _lib.ptr(vals.get(0)), _lib.ptr(vals.get(H - 1)), 0.0, 0.0, None,
                                              B, H, D, _lib.current_stream()), "mpdx_add_noise")
        return x
    import ctypes as C
    keys = list(vals)
    lib = _lib.load()
    for i in range(0, len(keys), 16):   # (the kernel takes 16 entries per launch; order is kept across launches)
        part = keys[i:i + 16]
        idx = (C.c_int32 * len(part))(*part)
        ptrs = (C.c_void_p * len(part))(*[vals[k].data_ptr() for k in part])
        _lib.check(lib.mpdx_hard_conds(x.data_ptr(), None, len(part), idx, ptrs, B, H, D, _lib.current_stream()), "mpdx_hard_conds")
    return x


def extract(a, t, x_shape):
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def step_coefs(model, t: int, noise_std_extra: float = 1.0, scale_grad_by_std: bool = False) -> "_lib.StepCoefs":
    """The t-th entries of the model's registered buffers as the scalar block mpdx_ddpm_step takes.
    scale_grad_by_std: guide increments of this step are multiplied by model_var[t] (sample_functions.py:41-43,77-78)."""
    c = model.host_buffers()
    scale = 0.0 if t == 0 else float(c["noise_scale"][t])  # noise[t == 0] = 0  (sample_functions.py:52)
    return _lib.StepCoefs(float(c["sqrt_recip_alphas_cumprod"][t]), float(c["sqrt_recipm1_alphas_cumprod"][t]),
                          float(c["posterior_mean_coef1"][t]), float(c["posterior_mean_coef2"][t]), scale,
                          float(noise_std_extra), int(bool(model.predict_epsilon)), int(bool(model.clip_denoised)), 1.0, 0.0,
                          float(c["model_var"][t]) if scale_grad_by_std else 1.0)


@torch.no_grad()
def ddpm_sample_fn(model, x, hard_conds, context, t, guide=None, n_guide_steps=1, scale_grad_by_std=False,
                   t_start_guide=torch.inf, noise_std_extra_schedule_fn=None, debug=False, noise=None, **kwargs):
    """One reverse step (sample_functions.py:17-62).  Returns (x_next, None); hard conditioning is applied by the
    caller afterwards, as p_sample_loop does (diffusion_model_base.py:173).  ``noise`` (optional) injects the
    randn_like draw for parity runs."""
    if context is not None:
        raise NotImplementedError("context is always None on this path (inference.py:182)")
    from .diffusion_model import timestep_hint
    t_single = timestep_hint(t)   # make_timesteps' tensors carry their value (no host sync); any other (or modified) tensor: read it, as the reference's `if t_single < 0`
    if t_single is None:
        t_single = int(t.reshape(-1)[0])
    tt = max(t_single, 0)
    B, H, D = x.shape
    x = x.to(torch.float32).contiguous()
    lib = _lib.load()
    hdl, packed, tab, ws = model.model.engine(model.n_diffusion_steps, B)
    extra = 1.0 if noise_std_extra_schedule_fn is None else float(noise_std_extra_schedule_fn(t_single))
    coefs = step_coefs(model, tt, extra)
    if noise is None and tt != 0:
        noise = torch.empty_like(x)
        model.fill_randn(noise)
    elif noise is not None:
        noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
    st = _lib.current_stream()
    use_guide = guide is not None and t_single < t_start_guide
    nz = _lib.ptr(noise) if tt != 0 else None
    out = x.clone()
    _lib.check(lib.mpdx_ddpm_step(hdl, packed.data_ptr(), tab.data_ptr(), model.model._timetab_T, out.data_ptr(),
                                  None if use_guide else nz, None, None, coefs, tt, 1 if use_guide else 0, None, None, B, B,
                                  ws.data_ptr(), st), "mpdx_ddpm_step")
    if use_guide:
        model_var = None
        if scale_grad_by_std:  # exp(posterior_log_variance_clipped[t]) (sample_functions.py:36); the host-side fp32 value, so that
            model_var = float(model.host_buffers()["model_var"][tt])  # this loop and the fused mpdx_plan multiply by the same bits
        out = guide_gradient_steps(out, hard_conds=hard_conds, guide=guide, n_guide_steps=n_guide_steps,
                                   scale_grad_by_std=scale_grad_by_std, model_var=model_var)
        _lib.check(lib.mpdx_add_noise(out.data_ptr(), nz, None, None, coefs.noise_scale, extra, None, B, H, D, st), "mpdx_add_noise")
    return out, None


def guide_gradient_steps(x, hard_conds=None, guide=None, n_guide_steps=1, scale_grad_by_std=False, model_var=None,
                         debug=False, **kwargs):
    """sample_functions.py:65-83."""
    for _ in range(int(n_guide_steps)):
        grad_scaled = guide(x)
        if scale_grad_by_std:
            grad_scaled = model_var * grad_scaled
        x = x + grad_scaled
        x = apply_hard_conditioning(x, hard_conds)
    return x
