"""GuideManagerTrajectoriesWithVelocity (oracle; test infrastructure).  mpd/models/diffusion_models/guides.py:149-236.

``cost(x, x_interpolated=..., return_invidual_costs_and_weights=True) -> (list of [B] tensors, list of floats)``
is the call-site contract of guides.py:190.  The gradient is taken w.r.t. the UNNORMALISED trajectory (x is rebound
at :180) and nevertheless added to the normalised one by the caller - reproduced, not fixed.
"""
import torch
import torch.nn.functional as F


def interpolate_points_v1(points: torch.Tensor, num_interpolated_points: int) -> torch.Tensor:
    """torch_robotics...distance_fields.interpolate_points_v1 (un-vendored; called at guides.py:184).
    PARITY UNPINNED: restated as linear interpolation with align_corners=True over the horizon axis."""
    p = points.transpose(-2, -1)
    p = F.interpolate(p, num_interpolated_points, mode="linear", align_corners=True)
    return p.transpose(-2, -1)


def clip_grad_by_norm(g: torch.Tensor, max_grad_norm: float = 1.0) -> torch.Tensor:
    # guides.py:224-230
    n = torch.linalg.norm(g + 1e-6, dim=-1, keepdim=True)
    return torch.clip(n, 0.0, max_grad_norm) / n * g


def clip_grad_by_value(g: torch.Tensor, max_grad_value: float = 0.1) -> torch.Tensor:
    # guides.py:232-236
    return torch.clip(g, -max_grad_value, max_grad_value)


class GuideManager:
    def __init__(self, normalizer, cost, clip_grad=True, max_grad_norm=1.0, interpolate=True, n_interp=128, clip_grad_rule="norm",
                 max_grad_value=0.1):
        self.normalizer, self.cost = normalizer, cost
        self.clip_grad, self.max_grad_norm = clip_grad, max_grad_norm
        self.clip_grad_rule, self.max_grad_value = clip_grad_rule, max_grad_value
        self.interpolate, self.n_interp = interpolate, n_interp

    def __call__(self, x_normalized: torch.Tensor) -> torch.Tensor:
        x = x_normalized.clone()
        with torch.enable_grad():
            x.requires_grad_(True)
            x = self.normalizer.unnormalize(x)  # :180 (rebinds x)
            x_interp = interpolate_points_v1(x, self.n_interp) if self.interpolate else x  # :182-186
            cost_l, w_l = self.cost(x, x_interpolated=x_interp, return_invidual_costs_and_weights=True)  # :190
            grad = 0
            for c, w in zip(cost_l, w_l):
                if torch.is_tensor(c):
                    g = torch.autograd.grad([c.sum()], [x], retain_graph=True)[0]  # :196
                    if self.clip_grad:  # guides.py:213-222
                        if self.clip_grad_rule == "norm":
                            g = clip_grad_by_norm(g, self.max_grad_norm)
                        elif self.clip_grad_rule == "value":
                            g = clip_grad_by_value(g, self.max_grad_value)
                        else:
                            raise NotImplementedError
                    g[..., 0, :] = 0.0  # :202-203
                    g[..., -1, :] = 0.0
                    grad = grad + w * g
        return -1.0 * grad  # :210
