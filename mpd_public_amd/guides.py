"""GuideManagerTrajectoriesWithVelocity - drop-in for mpd/models/diffusion_models/guides.py:149-236.

Same constructor (incl. the `**kwargs` that swallows inference.py:234's misspelt `num_interpolated_points`, so the
effective number of interpolated points stays the class default 128) and the same call protocol
``guide(x_normalized[B,H,D]) -> increment[B,H,D]`` (already negated and weighted).  The cost composite is compiled
once into `mpdx_guide_params`; every call is one launch of the HIP guide kernel (csrc/guide.hpp).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .planning import CostCollision, CostComposite, CostGPTrajectory


def build_device_params(robot, ws_dim, cutoff_margin, mins, maxs, cost_l, weight_l, interpolate, n_interp, clip_grad, max_grad_norm, device,
                        clip_grad_rule="norm", max_grad_value=0.1, identity_normalizer=False):
    """Compile cost descriptors into the `mpdx_guide_params` block the HIP kernels take.  Returns (params, primitive
    table tensor) - the caller keeps the tensor alive (params holds its raw device pointer)."""
    gp = _lib.GuideParams()
    gp.robot, gp.q_dim, gp.ws_dim = robot.robot_id, robot.q_dim, ws_dim
    gp.interpolate, gp.n_interp = int(bool(interpolate)), int(n_interp)
    gp.clip_grad, gp.max_grad_norm = int(bool(clip_grad)), float(max_grad_norm)
    if clip_grad_rule not in ("norm", "value"):
        raise NotImplementedError(f"clip_grad_rule={clip_grad_rule!r}")   # as guides.py:219-220
    gp.clip_rule, gp.max_grad_value = (1 if clip_grad_rule == "value" else 0), float(max_grad_value)
    gp.identity_normalizer = int(identity_normalizer)   # 0 limits, 1 Identity, 2 GaussianNormalizer (means in `mins`, stds in `maxs`)
    D = 2 * robot.q_dim
    if mins is not None:
        mins, maxs = torch.as_tensor(mins).cpu().numpy(), torch.as_tensor(maxs).cpu().numpy()
        for d in range(D):
            gp.mins[d], gp.maxs[d] = float(mins[d]), float(maxs[d])
    gp.cutoff_margin, gp.link_margin = float(cutoff_margin), float(robot.link_margin)
    prims, nf, off = [], 0, 0
    gp.use_gp = 0
    for c, w in zip(cost_l, weight_l):
        if isinstance(c, CostCollision):
            if nf >= _lib.MAX_FIELDS:
                raise NotImplementedError(f"at most {_lib.MAX_FIELDS} collision fields")
            f, fld = gp.fields[nf], c.field
            f.kind, f.weight = fld.kind, float(w)
            if fld.kind == _lib.FIELD_OBJECTS:
                sp, bx = fld.objects.prim_floats()
                f.sphere_off, f.n_spheres = off, sp.size // 4
                off += sp.size
                f.box_off, f.n_boxes = off, bx.size // 6
                off += bx.size
                prims += [sp, bx]
            elif fld.kind == _lib.FIELD_WORKSPACE:
                for j in range(ws_dim):
                    f.ws_min[j], f.ws_max[j] = float(fld.ws_min[j]), float(fld.ws_max[j])
            nf += 1
        elif isinstance(c, CostGPTrajectory):
            if gp.use_gp:
                raise NotImplementedError("one CostGPTrajectory term")
            gp.use_gp, gp.gp_weight, gp.dt, gp.sigma_gp = 1, float(w), float(c.dt), float(c.sigma_gp)
            gp.gp_half_factor = int(bool(getattr(c, "half_factor", False)))
        else:
            raise NotImplementedError(type(c))
    gp.n_fields = nf
    n_floats = int(sum(p.size for p in prims))
    table = np.concatenate(prims).astype(np.float32) if n_floats else np.zeros(4, np.float32)
    prim_t = torch.from_numpy(table).to(device)
    gp.prims, gp.n_prim_floats = prim_t.data_ptr(), n_floats
    return gp, prim_t


class GuideManagerTrajectoriesWithVelocity(nn.Module):
    def __init__(self, dataset, cost, clip_grad=False, clip_grad_rule="norm", max_grad_norm=1.0, max_grad_value=0.1,
                 interpolate_trajectories_for_collision=False, num_interpolated_points_for_collision=128,
                 start_state_pos=None, goal_state_pos=None, num_steps=100, robot=None, n_samples=1, tensor_args=None, **kwargs):
        super().__init__()
        # A CostComposite of the reference's cost terms is compiled into the HIP guide kernel (hand-derived gradients).
        # Any other callable with the call-site contract of guides.py:190,
        #     cost(x, x_interpolated=..., return_invidual_costs_and_weights=True) -> ([B] tensors, [weights]),
        # is differentiated with torch autograd ON THE GPU exactly as the reference's manager does (guides.py:173-211);
        # such a guide runs on the step-by-step protocol loop, not inside mpdx_plan.
        self.is_native = isinstance(cost, CostComposite)
        if not self.is_native and not callable(cost):
            raise TypeError("cost must be a CostComposite or a callable with the contract of guides.py:190")
        if clip_grad_rule not in ("norm", "value"):
            raise NotImplementedError(f"clip_grad_rule={clip_grad_rule!r}")   # as guides.py:219-220
        self.cost, self.dataset = cost, dataset
        self.interpolate_trajectories_for_collision = interpolate_trajectories_for_collision
        self.num_interpolated_points_for_collision = num_interpolated_points_for_collision
        self.clip_grad, self.clip_grad_rule = clip_grad, clip_grad_rule
        self.max_grad_norm, self.max_grad_value = max_grad_norm, max_grad_value
        self._params = None
        self._prims = None
        self._flag = None

    # ------------------------------------------------------------------------------------------- compile to device params
    def device_params(self, device) -> "_lib.GuideParams":
        if self._params is not None and self._prims.device == torch.device(device):
            return self._params
        ds = self.dataset
        if ds.state_dim != 2 * ds.robot.q_dim:
            raise NotImplementedError("the velocity guide needs include_velocity=True (state = pos + vel)")
        kind = getattr(ds.normalizer, "kind", "limits")
        if kind not in ("limits", "identity", "gaussian"):
            raise NotImplementedError(f"the HIP guide un-normalises with limits (LimitsNormalizer and its subclasses), mean / std (GaussianNormalizer) or not at "
                                      f"all (Identity); {type(ds.normalizer).__name__} is not supported under a guide")
        # the kernel's un-normalisation: 0 = limits with the whole-tensor range test (normalization.py:156-167), 1 = none (:111-116),
        # 2 = x * stds + means (:140-141; the two vectors travel in the `mins` / `maxs` slots, no range test)
        mode = {"limits": 0, "identity": 1, "gaussian": 2}[kind]
        lo = None if mode == 1 else ds.normalizer.means if mode == 2 else ds.normalizer.mins
        hi = None if mode == 1 else ds.normalizer.stds if mode == 2 else ds.normalizer.maxs
        self._params, self._prims = build_device_params(
            ds.robot, ds.env.dim, ds.task.obstacle_cutoff_margin, lo, hi,
            self.cost.cost_l, self.cost.weight_cost_l, self.interpolate_trajectories_for_collision, self.num_interpolated_points_for_collision,
            self.clip_grad, self.max_grad_norm, device, clip_grad_rule=self.clip_grad_rule, max_grad_value=self.max_grad_value,
            identity_normalizer=mode)
        return self._params

    # ------------------------------------------------------------------------------------------- guide protocol
    def _forward_autograd(self, x_normalized):
        """guides.py:173-211 verbatim in behaviour, for a Python cost callable: torch autograd on the device."""
        if not x_normalized.is_cuda:
            raise RuntimeError("the guide runs on the GPU; there is no CPU fallback")
        x = x_normalized.detach().clone()
        with torch.enable_grad():
            x.requires_grad_(True)
            x = self.dataset.unnormalize_trajectories(x)
            if self.interpolate_trajectories_for_collision:
                xi = torch.nn.functional.interpolate(x.transpose(-2, -1), self.num_interpolated_points_for_collision, mode="linear",
                                                     align_corners=True).transpose(-2, -1)
            else:
                xi = x
            cost_l, w_l = self.cost(x, x_interpolated=xi, return_invidual_costs_and_weights=True)
            grad = 0
            for c, w in zip(cost_l, w_l):
                if torch.is_tensor(c):
                    g = torch.autograd.grad([c.sum()], [x], retain_graph=True)[0]
                    if self.clip_grad:
                        if self.clip_grad_rule == "norm":
                            n = torch.linalg.norm(g + 1e-6, dim=-1, keepdims=True)
                            g = torch.clip(n, 0.0, self.max_grad_norm) / n * g
                        else:
                            g = torch.clip(g, -self.max_grad_value, self.max_grad_value)
                    g[..., 0, :] = 0.0
                    g[..., -1, :] = 0.0
                    grad = grad + w * g
        return (-1.0 * grad).detach()

    @torch.no_grad()
    def forward(self, x_normalized):
        if not self.is_native:
            return self._forward_autograd(x_normalized)
        x = x_normalized.to(torch.float32).contiguous()
        B, H, D = x.shape
        gp = self.device_params(x.device)
        lib, st = _lib.load(), _lib.current_stream()
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(lib.mpdx_absmax(x.data_ptr(), flag.data_ptr(), B, B, H, D, st), "mpdx_absmax")
        out = torch.empty_like(x)
        _lib.check(lib.mpdx_guide_step(C.byref(gp), x.data_ptr(), out.data_ptr(), None, None, flag.data_ptr(), None, B, B, H, D, st),
                   "mpdx_guide_step")
        return out
