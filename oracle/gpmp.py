"""GPMP2 (oracle; test infrastructure) - one Levenberg-Marquardt step of the trajectory optimiser the dataset-generation script
runs (scripts/generate_data/generate_trajectories.py:94-120: `GPMP2(**planner_params)` inside `HybridPlanner`).

PARITY UNPINNED: the planner lives in the un-vendored `mp_baselines` submodule (empty in /root/reference, no recoverable SHA) and the
reference holds no tests or vectors for it.  Restated from the published algorithm - Mukadam, Dong, Yan, Dellaert, Boots,
"Continuous-time Gaussian process motion planning via probabilistic inference", IJRR 2018, sections 4.1-4.3:

    theta* = argmin  1/2 || theta - mu ||^2_K  +  1/2 || h(theta) ||^2_{Sigma_obs}
    - GP prior with the constant-velocity model: factors e_i = theta_{i+1} - Phi(dt) theta_i, covariance Q = sigma_gp^2 Q_c(dt),
      Q^-1 = [[12/dt^3, -6/dt^2], [-6/dt^2, 4/dt]] (x) I; start and goal states FIXED (the sigma -> 0 limit of the paper's prior factors)
    - obstacle factors: hinge losses of every link sphere against every collision field (oracle/costs.py CostCollision.factors),
      evaluated on the interpolated trajectory (the same 128 points the guide's collision cost uses), Sigma_obs = sigma_obs^2 I
    - Levenberg-Marquardt on the Gauss-Newton normal equations:
          ( J^T J + lambda diag(J^T J) ) delta = - J^T r ,      r = the stacked whitened residuals, J = dr / d theta_free

Here J is torch forward-mode autograd and the solve is dense float64: the HIP kernel (csrc/planner.hpp) builds the same system
block-tridiagonal in LDS with hand-derived Jacobians.
"""
from __future__ import annotations

import torch

from .guide import interpolate_points_v1


def residuals(theta: torch.Tensor, robot, collision_costs, dt: float, sigma_gp: float, sigma_obs: float, n_interp: int) -> torch.Tensor:
    """theta [H, 2q] (one trajectory, raw units) -> the stacked whitened residual vector r with F = 1/2 r.r"""
    qd = robot.q_dim
    q, v = theta[:, :qd], theta[:, qd:]
    eq = q[1:] - q[:-1] - dt * v[:-1]
    ev = v[1:] - v[:-1]
    # Q^-1 = L L^T per joint with L = [[sqrt(12/dt^3), 0], [-6/dt^2 / sqrt(12/dt^3), sqrt(4/dt - 3/dt)]]:  e^T Q^-1 e = |L^T e|^2
    l11 = (12.0 / dt ** 3) ** 0.5
    l21 = (-6.0 / dt ** 2) / l11
    l22 = (4.0 / dt - l21 * l21) ** 0.5
    w1 = (l11 * eq + l21 * ev) / sigma_gp
    w2 = (l22 * ev) / sigma_gp
    xi = interpolate_points_v1(theta[None], n_interp) if n_interp else theta[None]
    obs = [c.factors(xi).reshape(-1) / sigma_obs for c in collision_costs]
    return torch.cat([w1.reshape(-1), w2.reshape(-1)] + obs)


def objective(theta, robot, collision_costs, dt, sigma_gp, sigma_obs, n_interp) -> torch.Tensor:
    r = residuals(theta, robot, collision_costs, dt, sigma_gp, sigma_obs, n_interp)
    return 0.5 * (r * r).sum()


def lm_step(theta: torch.Tensor, robot, collision_costs, dt: float, sigma_gp: float, sigma_obs: float, n_interp: int, lam: float):
    """One damped Gauss-Newton step for ONE trajectory theta [H, 2q] (float64 recommended).  Returns (delta [H, 2q] with zero rows at
    the fixed start / goal states, F(theta))."""
    H, D = theta.shape
    free0 = theta[1:-1].reshape(-1).detach().clone()

    def r_of(free):
        th = torch.cat([theta[:1], free.reshape(H - 2, D), theta[-1:]], dim=0)
        return residuals(th, robot, collision_costs, dt, sigma_gp, sigma_obs, n_interp)
    J = torch.func.jacfwd(r_of)(free0)            # [n_res, (H-2) D]
    r = r_of(free0)
    A = J.T @ J
    g = J.T @ r
    A = A + lam * torch.diag(torch.diagonal(A))
    delta = -torch.linalg.solve(A, g)
    out = torch.zeros_like(theta)
    out[1:-1] = delta.reshape(H - 2, D)
    return out, 0.5 * (r * r).sum()
