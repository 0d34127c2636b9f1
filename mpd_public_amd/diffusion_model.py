"""GaussianDiffusionModel - drop-in for mpd.models.diffusion_models.diffusion_model_base.GaussianDiffusionModel
(diffusion_model_base.py:46-316): same constructor, same registered buffers (state-dict compatible), same
``run_inference / conditional_sample / p_sample_loop / p_mean_variance / warmup`` protocol.

The sampling half (DDPM and DDIM) runs on libmpdx.so.  The training half (:320-357): q_sample / p_losses return the forward
value (per-sample timesteps, hard conditioning, WeightedL1/L2) without autograd history; `loss()` - what the reference's trainer
calls - returns a loss whose backward() is the native backward pass (mpd_public_amd/trainer.py, csrc/train.hpp).
"""
from __future__ import annotations

from copy import copy

import torch
import torch.nn as nn

from . import _lib
from .sample_functions import apply_hard_conditioning, ddpm_sample_fn, extract, step_coefs
from .schedules import diffusion_buffers


def make_timesteps(batch_size, i, device):
    t = torch.full((batch_size,), i, device=device, dtype=torch.long)
    t._mpdx_value = int(i)   # the loop index rides on the tensor: this package's sample functions read it instead of synchronising (`int(t[0])`, sample_functions.py:28)
    t._mpdx_version = t._version   # ... while nobody has written to the tensor since (an in-place `t -= 1` bumps the version counter: the hint is dropped)
    return t


def timestep_hint(t):
    """The batch-constant value of a timestep tensor made by make_timesteps and not modified since, else None (the caller reads the tensor: one host sync)."""
    v = getattr(t, "_mpdx_value", None)
    return v if v is not None and getattr(t, "_mpdx_version", -1) == t._version else None


class GaussianDiffusionModel(nn.Module):
    def __init__(self, model=None, variance_schedule="exponential", n_diffusion_steps=100, clip_denoised=True,
                 predict_epsilon=False, loss_type="l2", context_model=None, **kwargs):
        super().__init__()
        self.model = model
        self.context_model = context_model
        self.n_diffusion_steps = n_diffusion_steps
        self.state_dim = self.model.state_dim
        self.clip_denoised = clip_denoised
        self.predict_epsilon = predict_epsilon
        self.loss_type = loss_type
        self.loss_fn = {"l1": "WeightedL1", "l2": "WeightedL2"}.get(loss_type, loss_type)  # the reference keeps the loss module here
        bufs = diffusion_buffers(variance_schedule, n_diffusion_steps)
        if not all(bool(torch.isfinite(v).all()) for v in bufs.values()):
            # the reference's exponential_beta_schedule (helpers.py:40-46) rounds beta_{T-1} ABOVE 1 for most step counts (finite for T = 1, 25, 47, 50, 55,
            # 61, 73, 94, 97, 100, ... - the shipped models use 25 and 100): alphas turn negative and the square roots NaN.  Reproduced bit for bit, like
            # every other buffer - but said out loud, because every plan of such a model is NaN (in the reference too)
            import warnings
            warnings.warn(f"variance_schedule={variance_schedule!r} with n_diffusion_steps={n_diffusion_steps} yields non-finite schedule buffers (the reference's "
                          "formula does: its last beta rounds above 1) - sampling and training will produce NaN; use 25 / 50 / 100 steps or the cosine schedule",
                          RuntimeWarning, stacklevel=2)
        for name, value in bufs.items():
            self.register_buffer(name, value)
        self._host = None
        self._host_stamp = None
        self._coef_cache = {}
        self._rng_seed = 0
        self._rng_offset = 0
        self.in_kernel_noise_min_bytes = 64 << 20   # plans whose noise stream is at least this large draw it inside the step kernels

    # ---------------------------------------------------------------------------------------------- helpers
    def host_buffers(self):
        """CPU copies of the schedule buffers (scalars are passed to the kernels by value)."""
        d = self.__dict__   # (the buffer list is cached like TemporalUnet._param_stamp's parameter list: the walk cost 0.1 ms per denoising step of the protocol loop)
        n = d.get("_hb_calls", 0) + 1
        d["_hb_calls"] = n
        bl = d.get("_blist")
        if bl is None or n % 16 == 0:
            bl = d["_blist"] = [b for k, b in self.named_buffers() if "." not in k]
        stamp = tuple((b.data_ptr(), b._version) for b in bl)
        if self._host is None or self._host_stamp != stamp:
            host = {k: v.detach().to("cpu", torch.float32) for k, v in self.named_buffers() if "." not in k}
            # model_std = exp(0.5 * posterior_log_variance_clipped[t])  (sample_functions.py:35-36), fp32 like the reference
            host["noise_scale"] = torch.exp(0.5 * host["posterior_log_variance_clipped"])
            host["model_var"] = torch.exp(host["posterior_log_variance_clipped"])   # sample_functions.py:36 (scale_grad_by_std)
            self._host = {k: v.numpy() for k, v in host.items()}
            self._host_stamp = stamp
            self._coef_cache = {}
        return self._host

    def _apply(self, fn, *a, **k):
        self.__dict__["_blist"] = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.__dict__["_blist"] = None
        return r

    def manual_seed(self, seed: int):
        """Seed of the device noise generator (Philox counter stream of mpdx_randn)."""
        self._rng_seed, self._rng_offset = int(seed), 0
        return self

    def fill_randn(self, out: torch.Tensor):
        n = out.numel()
        _lib.check(_lib.load().mpdx_randn(out.data_ptr(), n, self._rng_seed, self._rng_offset, _lib.current_stream()), "mpdx_randn")
        self._rng_offset += (n + 3) // 4
        return out

    # ---------------------------------------------------------------------------------------------- fused loop
    def _coef_table(self, noise_std_extra_schedule_fn, scale_grad_by_std=False):
        """ctypes array [T] of mpdx_step_coefs, cached per (schedule buffers, extra-noise schedule values, guide scaling)."""
        T = self.n_diffusion_steps
        self.host_buffers()
        extras = tuple(1.0 if noise_std_extra_schedule_fn is None else float(noise_std_extra_schedule_fn(t)) for t in range(T))
        key = (extras, bool(scale_grad_by_std))
        arr = self._coef_cache.get(key)
        if arr is None:
            arr = (_lib.StepCoefs * T)()
            for t in range(T):
                arr[t] = step_coefs(self, t, extras[t], scale_grad_by_std)
            self._coef_cache[key] = arr
        return arr

    @torch.no_grad()
    def plan(self, hard_conds, n_samples, horizon=None, n_diffusion_steps_without_noise=0, noise=None,
             noise_std_extra_schedule_fn=None, return_chain=True, guide=None, n_guide_steps=1, t_start_guide=float("inf"),
             n_per_context=None, scale_grad_by_std=False):
        """The whole reverse loop of p_sample_loop (diffusion_model_base.py:157-182) enqueued by ONE mpdx_plan call,
        without host synchronisation.  hard_conds: {0: start[B,D] or [D], H-1: goal}.  Returns (x_final, chain or None)
        with chain laid out [steps+1, B, H, D] (run_inference's order).
        guide: a mpd_public_amd.GuideManagerTrajectoriesWithVelocity (device guide) or None.
        n_per_context: trajectories per start/goal context when hard_conds are per-trajectory [B,D] tables of several
        contexts (the whole-tensor range test of LimitsNormalizer is evaluated per context, as one reference call per
        context would)."""
        H = horizon or self.model.n_support_points
        D, T, n0, B = self.state_dim, self.n_diffusion_steps, int(n_diffusion_steps_without_noise), int(n_samples)
        dev = self.betas.device
        if set(hard_conds.keys()) - {0, H - 1}:
            raise NotImplementedError("fused plan supports hard conditions at horizon indices 0 and H-1 only")

        def cond(v):
            if v is None:
                return None
            v = v.to(device=dev, dtype=torch.float32)
            return (v.reshape(1, -1).expand(B, -1) if v.dim() == 1 else v).contiguous()

        hs, hg = cond(hard_conds.get(0)), cond(hard_conds.get(H - 1))
        hdl, packed, tab, ws = self.model.engine(T, B)
        steps = T + n0
        n = B * H * D
        rng_seed, rng_offset, noise_ptr = 0, 0, None
        if noise is None:
            # production path: x_T from the device generator, every step's draw generated IN the step kernels from the same
            # Philox stream (no [steps+1, B, H, D] noise tensor: 2.4 GB for a 6400-trajectory Panda shard).  Element i of the
            # stream is what fill_randn of one big tensor would have put at flat index i, so both routes give the same bits.
            if n % 4:
                raise ValueError("B*H*D must be a multiple of 4 (one Philox counter yields 4 normals)")
            if (steps + 1) * n * 4 < self.in_kernel_noise_min_bytes:
                # small plans: one generator launch for the whole stream is cheaper than a Philox evaluation per element inside
                # the step kernels (measured at B=100: 24.99 vs 25.24 ms per plan) - the bits are the same either way
                noise = self.fill_randn(torch.empty((steps + 1, B, H, D), device=dev, dtype=torch.float32))
                x = noise[0].clone()
                noise_ptr = noise[1:].data_ptr()
            else:
                rng_seed, rng_offset = self._rng_seed, self._rng_offset
                x = torch.empty((B, H, D), device=dev, dtype=torch.float32)
                _lib.check(_lib.load().mpdx_randn(x.data_ptr(), n, rng_seed, rng_offset, _lib.current_stream()), "mpdx_randn")
                self._rng_offset += (steps + 1) * n // 4
        else:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(noise.shape) == (steps + 1, B, H, D), noise.shape
            x = noise[0].clone()
            noise_ptr = noise[1:].data_ptr()
        coefs = self._coef_table(noise_std_extra_schedule_fn, scale_grad_by_std)
        chain = torch.empty((steps + 1, B, H, D), device=dev, dtype=torch.float32) if return_chain else None
        npc = int(n_per_context or B)
        gp_ref, flags, n_gs, t_sg = None, None, 0, 0
        if int(n_guide_steps) <= 0:
            guide = None   # range(0): the reference runs no guide iteration
        if guide is not None:
            import ctypes as C
            gp_ref = C.byref(guide.device_params(dev))
            n_gs = int(n_guide_steps)
            t_sg = int(min(t_start_guide, T + 1)) if t_start_guide != float("inf") else T + 1
            flags = torch.empty(steps * (n_gs + 1) * ((B + npc - 1) // npc), dtype=torch.int32, device=dev)
        # T here is the LOOP length / coefficient-table length (not the time-table capacity)
        _lib.check(_lib.load().mpdx_plan(hdl, packed.data_ptr(), tab.data_ptr(), T, coefs, n0,
                                         x.data_ptr(), noise_ptr, _lib.ptr(hs), _lib.ptr(hg), _lib.ptr(chain), B,
                                         ws.data_ptr(), gp_ref, n_gs, t_sg, _lib.ptr(flags), npc, rng_seed, rng_offset,
                                         _lib.current_stream()), "mpdx_plan")
        return x, chain

    # ---------------------------------------------------------------------------------------------- sampling
    # The three elementwise helpers of the reference's class (diffusion_model_base.py:109-141), kept for callers that use them directly: plain tensor
    # arithmetic on the registered schedule buffers (any device).  The planning loop itself never calls them - its U-Net pass, x0 prediction, clamp and
    # posterior mean are ONE fused kernel sequence (p_mean_variance below / mpdx_plan).
    def predict_noise_from_start(self, x_t, t, x0):
        if self.predict_epsilon:   # the network output already is the noise
            return x0
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0) / extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def predict_start_from_noise(self, x_t, t, noise):
        if not self.predict_epsilon:   # the network output already is x0
            return noise
        return extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise

    def q_posterior(self, x_start, x_t, t):
        mean = extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def p_mean_variance(self, x, hard_conds, context, t):
        """(model_mean, posterior_variance, posterior_log_variance) as diffusion_model_base.py:143-155."""
        if context is not None:
            raise NotImplementedError("context is always None on this path")
        if not self.clip_denoised:
            raise RuntimeError("clip_denoised=False is an error in the reference too (:152)")
        tt = timestep_hint(t)
        if tt is None:
            tt = int(t.reshape(-1)[0])
        B = x.shape[0]
        hdl, packed, tab, ws = self.model.engine(self.n_diffusion_steps, B)
        mean = x.to(torch.float32).contiguous().clone()
        _lib.check(_lib.load().mpdx_ddpm_step(hdl, packed.data_ptr(), tab.data_ptr(), self.model._timetab_T, mean.data_ptr(), None,
                                              None, None, step_coefs(self, tt), tt, 1, None, None, B, B, ws.data_ptr(),
                                              _lib.current_stream()), "mpdx_ddpm_step")
        return mean, extract(self.posterior_variance, t, x.shape), extract(self.posterior_log_variance_clipped, t, x.shape)

    def _normalise_hard_conds(self, hard_conds, B, device):
        """{t: [D] or [B,D] on any device} -> {t: contiguous fp32 [B,D] on `device`} (what the kernels index as hs[b*D+d];
        the reference's `x[:, t, :] = val` broadcasts a [D] value)."""
        out = {}
        for k, v in (hard_conds or {}).items():
            v = torch.as_tensor(v).to(device=device, dtype=torch.float32)
            if v.dim() == 1:
                v = v.reshape(1, -1).expand(B, -1)
            if v.shape != (B, self.state_dim):
                raise ValueError(f"hard condition at horizon index {k}: expected [{self.state_dim}] or [{B},{self.state_dim}], got {tuple(v.shape)}")
            out[k] = v.contiguous()
        return out

    @torch.no_grad()
    def p_sample_loop(self, shape, hard_conds, context=None, return_chain=False, sample_fn=ddpm_sample_fn,
                      n_diffusion_steps_without_noise=0, noise=None, **sample_kwargs):
        """diffusion_model_base.py:157-182.  ``noise`` ([steps+1, B, H, D], optional) injects the random stream."""
        device = self.betas.device
        batch_size = shape[0]
        if noise is not None:
            noise = noise.to(device=device, dtype=torch.float32).contiguous()   # converted once: noise[k] is handed to the kernels
            x = noise[0].clone()
        else:
            x = self.fill_randn(torch.empty(shape, device=device, dtype=torch.float32))
        hard_conds = self._normalise_hard_conds(hard_conds, batch_size, device)
        x = apply_hard_conditioning(x, hard_conds)
        chain = [x] if return_chain else None
        k = 1
        # the weights cannot change inside one loop: they are compared with the engine's pack ONCE here, not in every step's eps-model call
        unet = self.model
        freeze = hasattr(unet, "engine") and next(unet.parameters()).device.type == "cuda"
        if freeze:
            unet.engine(self.n_diffusion_steps, batch_size)
            unet.__dict__["_weights_frozen"] = True
        try:
            for i in reversed(range(-n_diffusion_steps_without_noise, self.n_diffusion_steps)):
                t = make_timesteps(batch_size, i, device)
                if noise is not None:
                    sample_kwargs["noise"] = noise[k]
                x, values = sample_fn(self, x, hard_conds, context, t, **sample_kwargs)
                x = apply_hard_conditioning(x, hard_conds)
                if return_chain:
                    chain.append(x)
                k += 1
        finally:
            if freeze:
                unet.__dict__["_weights_frozen"] = False
        if return_chain:
            return x, torch.stack(chain, dim=1)
        return x

    @torch.no_grad()
    def ddim_sample(self, shape, hard_conds, context=None, return_chain=False, t_start_guide=torch.inf, guide=None, n_guide_steps=1,
                    noise=None, **sample_kwargs):
        """DDIM sampler, drop-in for diffusion_model_base.py:184-259 (sampling_timesteps = T // 5, eta = 0): per time pair
        one U-Net pass + DDIM update (mpdx_ddpm_step mode 2), optional guide steps when t_next < t_start_guide, hard
        conditioning.  `noise` ([1+pairs, B, H, D], optional) injects x_T for parity runs (eta = 0: no other draw matters)."""
        if context is not None:
            raise NotImplementedError("context is always None on this path")
        device = self.betas.device
        B, H, D = shape
        total, sampling = self.n_diffusion_steps, self.n_diffusion_steps // 5
        times = torch.linspace(0, total - 1, steps=sampling + 1)
        times = list(reversed(torch.cat((torch.tensor([-1.0]), times)).int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        if noise is not None:
            x = noise[0].to(device=device, dtype=torch.float32).clone()
        else:
            x = self.fill_randn(torch.empty(shape, device=device, dtype=torch.float32))
        hard_conds = self._normalise_hard_conds(hard_conds, B, device)   # [D] values broadcast, CPU values moved: the kernel reads hs[b*D+d]
        x = apply_hard_conditioning(x, hard_conds)
        chain = [x.clone()] if return_chain else None
        hdl, packed, tab, ws = self.model.engine(total, B)
        lib, hb = _lib.load(), self.host_buffers()
        keys = set(hard_conds.keys())
        native_hc = keys <= {0, H - 1}
        hs = hard_conds.get(0) if native_hc else None
        hg = hard_conds.get(H - 1) if native_hc else None
        for time, time_next in pairs:
            guided = guide is not None and time_next >= 0 and time_next < t_start_guide
            c = step_coefs(self, time)
            if time_next < 0:
                c.ddim_k1, c.ddim_k2 = 1.0, 0.0          # x = x_start  (:221-226)
            else:
                a_next = torch.tensor(float(hb["alphas_cumprod"][time_next]), dtype=torch.float32)
                c.ddim_k1 = float(a_next.sqrt())
                c.ddim_k2 = float((1 - a_next - torch.tensor(0.0) ** 2).sqrt())   # eta = 0 -> sigma = 0 (:231-234)
            in_kernel_hc = native_hc and not guided
            _lib.check(lib.mpdx_ddpm_step(hdl, packed.data_ptr(), tab.data_ptr(), self.model._timetab_T, x.data_ptr(), None,
                                          _lib.ptr(hs) if in_kernel_hc else None, _lib.ptr(hg) if in_kernel_hc else None, c, time, 2,
                                          None, None, B, B, ws.data_ptr(), _lib.current_stream()), "mpdx_ddpm_step(ddim)")
            if guided:
                from .sample_functions import guide_gradient_steps
                # the reference names n_guide_steps in ddim_sample's signature and never forwards it (:240-246): ONE guide step per
                # time pair whatever the caller passed - reproduced
                x = guide_gradient_steps(x, hard_conds=hard_conds, guide=guide, **sample_kwargs)
            if not in_kernel_hc:
                x = apply_hard_conditioning(x, hard_conds)
            if return_chain:
                chain.append(x.clone())
        if return_chain:
            return x, torch.stack(chain, dim=1)
        return x

    @torch.no_grad()
    def conditional_sample(self, hard_conds, horizon=None, batch_size=1, ddim=False, **sample_kwargs):
        horizon = horizon or self.horizon
        shape = (batch_size, horizon, self.state_dim)
        if ddim:
            return self.ddim_sample(shape, hard_conds, **sample_kwargs)
        return self.p_sample_loop(shape, hard_conds, **sample_kwargs)

    def forward(self, cond, *args, **kwargs):
        raise NotImplementedError  # as diffusion_model_base.py:274-276

    @torch.no_grad()
    def warmup(self, horizon=64, device="cuda"):
        x = torch.randn((2, horizon, self.state_dim), device=device)
        self.model(x, make_timesteps(2, 1, device), context=None)

    @torch.no_grad()
    def run_inference(self, context=None, hard_conds=None, n_samples=1, return_chain=False, **diffusion_kwargs):
        """diffusion_model_base.py:285-316: returns the chain [steps+1, n_samples, H, D] (or its last element)."""
        hard_conds = copy(hard_conds)
        if context is not None:
            raise NotImplementedError("context is always None on this path (inference.py:182)")
        kw = dict(diffusion_kwargs)
        horizon = kw.pop("horizon", None)
        fused = diffusion_kwargs.pop("fused", True)  # extension: fused=False forces the step-by-step protocol loop
        kw.pop("fused", None)
        from .guides import GuideManagerTrajectoriesWithVelocity as _NativeGuide
        g = kw.get("guide")
        native_guide = g is None or (isinstance(g, _NativeGuide) and g.is_native)
        if fused and kw.pop("sample_fn", ddpm_sample_fn) is ddpm_sample_fn and native_guide and not kw.get("ddim", False) \
                and set(hard_conds.keys()) <= {0, (horizon or self.model.n_support_points) - 1}:
            # fused path: one mpdx_plan call for the whole loop.  What stays on the step-by-step protocol loop below (same
            # kernels, one call per step): a caller-supplied sample_fn, DDIM, a guide around an arbitrary Python cost
            # (torch autograd), hard conditions at horizon indices other than 0 / H-1.
            x, chain = self.plan(hard_conds, n_samples, horizon, kw.get("n_diffusion_steps_without_noise", 0), kw.get("noise"),
                                 kw.get("noise_std_extra_schedule_fn"), return_chain=True, guide=g,
                                 n_guide_steps=kw.get("n_guide_steps", 1), t_start_guide=kw.get("t_start_guide", float("inf")),
                                 scale_grad_by_std=bool(kw.get("scale_grad_by_std", False)))
            return chain if return_chain else chain[-1]
        for k, v in hard_conds.items():
            if v.dim() == 1:
                hard_conds[k] = v.reshape(1, -1).expand(n_samples, -1).contiguous()  # 'd -> b d'
        samples, chain = self.conditional_sample(hard_conds, context=None, batch_size=n_samples, return_chain=True,
                                                 **diffusion_kwargs)
        chain = chain.permute(1, 0, 2, 3)  # 'b diffsteps h d -> diffsteps b h d'
        if return_chain:
            return chain
        return chain[-1]

    # ---------------------------------------------------------------------------------------------- training half (diffusion_model_base.py:320-357)
    # q_sample / p_losses: forward values; loss(): with gradients enabled, the native forward + backward pass (trainer.py, csrc/train.hpp)
    def _hard_tables(self, hard_conds, B, H, D, device, allow_other=False):
        """{0: v, H-1: v} (v [D] or [B,D]) -> ([B,D] start table or None, [B,D] goal table or None): the two indices the kernels fold into their
        epilogues (what TrajectoryDataset.get_hard_conditions produces, trajectories.py:214-237).  Any other set of horizon indices: with
        `allow_other` (the forward-only q_sample / p_losses) -> (None, None) and the caller applies the whole dict with apply_hard_conditioning
        (mpdx_hard_conds); the native TRAINING pass takes 0 / H-1 only and refuses."""
        extra = set(int(k) for k in (hard_conds or {}).keys()) - {0, H - 1}
        if extra:
            if allow_other:
                return [None, None]
            raise NotImplementedError(f"hard conditions at horizon indices {sorted(extra)}: the native training pass supports 0 and H-1 only "
                                      "(sampling, q_sample and p_losses take any index)")
        out = []
        for key in (0, H - 1):
            v = (hard_conds or {}).get(key)
            if v is None:
                out.append(None)
                continue
            v = v.to(device=device, dtype=torch.float32)
            out.append((v.expand(B, D) if v.dim() == 1 else v).contiguous())
        return out

    @staticmethod
    def _other_indices(hard_conds, H):
        return bool(set(int(k) for k in (hard_conds or {}).keys()) - {0, H - 1})

    def q_sample(self, x_start, t, noise=None, hard_conds=None):
        """diffusion_model_base.py:320-330 (per-sample t).  `hard_conds` optionally folds the apply_hard_conditioning of
        p_losses (:335) into the same kernel.  Noise defaults to the device generator (mpdx_randn)."""
        if not x_start.is_cuda:
            raise RuntimeError("q_sample / p_losses run on the GPU (libmpdx); there is no CPU fallback")
        x_start = x_start.to(torch.float32).contiguous()
        B, H, D = x_start.shape
        if noise is None:
            noise = self.fill_randn(torch.empty((B, H, D), device=x_start.device, dtype=torch.float32))
        noise = noise.to(torch.float32).contiguous()
        t = t.to(device=x_start.device, dtype=torch.long).reshape(-1).contiguous()
        if t.numel() != B:
            raise ValueError(f"t must have {B} entries")
        hs, hg = self._hard_tables(hard_conds, B, H, D, x_start.device, allow_other=True)
        out = torch.empty_like(x_start)
        _lib.check(_lib.load().mpdx_q_sample(x_start.data_ptr(), noise.data_ptr(), t.data_ptr(), self.sqrt_alphas_cumprod.data_ptr(),
                                             self.sqrt_one_minus_alphas_cumprod.data_ptr(), hs.data_ptr() if hs is not None else None,
                                             hg.data_ptr() if hg is not None else None, out.data_ptr(), B, H, D, self.n_diffusion_steps,
                                             _lib.current_stream()), "mpdx_q_sample")
        if self._other_indices(hard_conds, H):
            apply_hard_conditioning(out, hard_conds)   # any horizon index (mpdx_hard_conds)
        return out

    def p_losses(self, x_start, context, t, hard_conds, noise=None):
        """diffusion_model_base.py:331-352, forward value only: (loss, info) with loss a 0-dim tensor WITHOUT autograd history
        (what the reference's validation pass computes under no_grad; `loss()` below is the entry with the native backward pass)."""
        if context is not None:
            raise NotImplementedError("context is always None on this path")
        if not x_start.is_cuda:
            raise RuntimeError("q_sample / p_losses run on the GPU (libmpdx); there is no CPU fallback")
        x_start = x_start.to(torch.float32).contiguous()
        B, H, D = x_start.shape
        if noise is None:
            noise = self.fill_randn(torch.empty((B, H, D), device=x_start.device, dtype=torch.float32))
        noise = noise.to(torch.float32).contiguous()
        x_noisy = self.q_sample(x_start, t, noise, hard_conds)
        x_recon = self.model(x_noisy, t, None)
        hs, hg = self._hard_tables(hard_conds, B, H, D, x_start.device, allow_other=True)
        if self._other_indices(hard_conds, H):
            x_recon = apply_hard_conditioning(x_recon, hard_conds)   # :343 for any horizon index (x_recon is this call's own tensor)
        target = noise if self.predict_epsilon else x_start
        if self.loss_type not in ("l1", "l2"):
            raise NotImplementedError(self.loss_type)
        out = torch.empty(1, dtype=torch.float32, device=x_start.device)
        _lib.check(_lib.load().mpdx_weighted_loss(x_recon.data_ptr(), target.data_ptr(), None, hs.data_ptr() if hs is not None else None,
                                                  hg.data_ptr() if hg is not None else None, 1 if self.loss_type == "l1" else 0, out.data_ptr(),
                                                  B, H, D, _lib.current_stream()), "mpdx_weighted_loss")
        return out[0], {}

    def loss(self, x, context, *args):
        """diffusion_model_base.py:354-357: uniform random timestep per sample, then p_losses.  With gradients enabled and a
        trainable U-Net the returned loss carries autograd history (trainer.loss_with_grad: the native forward + backward pass of
        csrc/train.hpp), so `loss.backward()` fills p.grad as in the reference's training loop; under torch.no_grad() (validation)
        it is the forward value only."""
        t = torch.randint(0, self.n_diffusion_steps, (x.shape[0],), device=x.device).long()
        if torch.is_grad_enabled() and context is None and any(p.requires_grad for p in self.model.parameters()):
            from .trainer import loss_with_grad
            hard_conds = args[0] if args else None
            return loss_with_grad(self, x, hard_conds, t=t), {}
        return self.p_losses(x, context, t, *args)
