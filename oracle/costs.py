"""Planning costs behind the guide (oracle; test infrastructure).  PARITY UNPINNED.

The reference builds these from un-vendored submodules that are EMPTY in /root/reference:
    mp_baselines.planners.costs.cost_functions.{CostCollision, CostGPTrajectory, CostComposite}  (inference.py:14,195-225)
    torch_robotics collision fields from task.get_collision_fields()                                (inference.py:191-193)
    torch_robotics robots (RobotPointMass, RobotPanda: differentiable FK + link collision spheres)
No pinned SHA is recoverable (.gitmodules:1-12, no .git), and the reference holds no tests or vectors for them.
This file restates the PUBLISHED formulas at the call-site contract
    cost(x, x_interpolated=..., return_invidual_costs_and_weights=True) -> ([B] per cost term, [weight per term])   (guides.py:190)
in plain differentiable torch, so that oracle/guide.py can take autograd gradients exactly as the reference's
guide manager does (guides.py:192-196).  The HIP kernels implement the same formulas with hand-derived gradients;
tests compare the two and check both against finite differences and closed forms.

Formulas
  * CostCollision (CHOMP/GPMP hinge on a signed-distance field, sigma_coll = 1):
        c = sum_{points i} sum_{link spheres k} relu( (r_k + cutoff_margin) - sdf(P_k(q_i)) )
    evaluated on the 128-point linear interpolation of the POSITION part of the trajectory.
      - objects field: sdf = min over primitives; sphere: |p-c| - r ; box: min(max_j d_j, 0) + |relu(d)|, d = |p-c| - h
      - workspace-boundary field: one hinge per face, sd = p_j - ws_min_j and ws_max_j - p_j
      - self-collision field (Panda): hinge on pair distances, relu(r_a + r_b - |P_a - P_b|)
  * CostGPTrajectory (GPMP2 constant-velocity GP prior, sigma_gp = 1, no 1/2 factor):
        e_i = x_{i+1} - Phi x_i,  Phi = [[I, dt I],[0, I]],  c = sum_i e_i^T Qinv e_i,
        Qinv = [[12/dt^3 I, -6/dt^2 I], [-6/dt^2 I, 4/dt I]]
    evaluated on the 64 support points (positions and velocities).
  * RobotPanda forward kinematics: Franka's published modified-DH parameters (Craig convention).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

# ----------------------------------------------------------------------------------------------------------- SDFs


def sdf_spheres(p: torch.Tensor, centers: torch.Tensor, radii: torch.Tensor) -> torch.Tensor:
    """p [...,dim], centers [n,dim], radii [n] -> [..., n]"""
    return torch.linalg.norm(p.unsqueeze(-2) - centers, dim=-1) - radii


def sdf_boxes(p: torch.Tensor, centers: torch.Tensor, half: torch.Tensor) -> torch.Tensor:
    d = (p.unsqueeze(-2) - centers).abs() - half
    inside = torch.minimum(d.amax(-1), torch.zeros_like(d[..., 0]))
    return inside + torch.linalg.norm(torch.relu(d), dim=-1)


@dataclass
class ObjectField:
    """signed distance to a set of sphere and box primitives (min over primitives)."""
    sphere_centers: torch.Tensor  # [ns, dim]
    sphere_radii: torch.Tensor    # [ns]
    box_centers: torch.Tensor     # [nb, dim]
    box_half: torch.Tensor        # [nb, dim]
    kind: str = "objects"

    def sdf(self, p):
        parts = []
        if self.sphere_radii.numel():
            parts.append(sdf_spheres(p, self.sphere_centers, self.sphere_radii))
        if self.box_centers.numel():
            parts.append(sdf_boxes(p, self.box_centers, self.box_half))
        return torch.cat(parts, dim=-1).min(-1)[0]


@dataclass
class WorkspaceField:
    ws_min: torch.Tensor
    ws_max: torch.Tensor
    kind: str = "workspace"


@dataclass
class SelfField:
    pairs: torch.Tensor  # [np, 2] indices into the robot's link spheres
    kind: str = "self"


# ----------------------------------------------------------------------------------------------------------- robots

PANDA_A = (0.0, 0.0, 0.0, 0.0825, -0.0825, 0.0, 0.088)
PANDA_D = (0.333, 0.0, 0.316, 0.0, 0.384, 0.0, 0.0)
PANDA_ALPHA = (0.0, -math.pi / 2, math.pi / 2, math.pi / 2, -math.pi / 2, math.pi / 2, math.pi / 2)
# collision spheres: (frame index 1..7, offset along that frame's z axis, radius)  - synthetic geometry
PANDA_SPHERES = ((1, -0.15, 0.10), (1, 0.0, 0.10), (3, -0.15, 0.09), (3, 0.0, 0.09), (4, 0.0, 0.09), (5, -0.25, 0.08),
                 (5, -0.12, 0.08), (5, 0.0, 0.08), (7, 0.0, 0.07), (7, 0.107, 0.06), (7, 0.17, 0.06))
# self-collision pairs (indices into PANDA_SPHERES): hand / flange / wrist against the base and upper-arm spheres
PANDA_SELF_PAIRS = ((8, 0), (8, 1), (8, 2), (9, 0), (9, 1), (9, 2), (9, 3), (10, 0), (10, 1), (10, 2), (10, 3), (10, 4))


def _mdh(alpha, a, d, theta):
    """Craig modified DH: Rot_x(alpha) Trans_x(a) Rot_z(theta) Trans_z(d); theta [...]. Returns [...,4,4]."""
    ca, sa = math.cos(alpha), math.sin(alpha)
    ct, st = torch.cos(theta), torch.sin(theta)
    z, o = torch.zeros_like(ct), torch.ones_like(ct)
    rows = [torch.stack([ct, -st, z, a * o], -1),
            torch.stack([st * ca, ct * ca, -sa * o, -sa * d * o], -1),
            torch.stack([st * sa, ct * sa, ca * o, ca * d * o], -1),
            torch.stack([z, z, z, o], -1)]
    return torch.stack(rows, -2)


class RobotPointMass:
    def __init__(self, q_dim=2, link_margin=0.01):
        self.q_dim, self.name = q_dim, "RobotPointMass"
        self.radii = torch.tensor([link_margin], dtype=torch.float32)

    def link_points(self, q):
        """q [..., q_dim] -> collision points [..., K, dim]"""
        return q.unsqueeze(-2)


class RobotPanda:
    def __init__(self):
        self.q_dim, self.name = 7, "RobotPanda"
        self.radii = torch.tensor([s[2] for s in PANDA_SPHERES], dtype=torch.float32)

    def frames(self, q):
        T = None
        out = []
        for i in range(7):
            Ti = _mdh(PANDA_ALPHA[i], PANDA_A[i], PANDA_D[i], q[..., i]).to(q.dtype)
            T = Ti if T is None else T @ Ti
            out.append(T)
        return out

    def link_points(self, q):
        fr = self.frames(q)
        pts = []
        for (k, off, _r) in PANDA_SPHERES:
            T = fr[k - 1]
            pts.append(T[..., :3, 3] + off * T[..., :3, 2])
        return torch.stack(pts, dim=-2)


# ----------------------------------------------------------------------------------------------------------- costs


class CostCollision:
    def __init__(self, robot, n_support_points, field=None, sigma_coll=1.0, cutoff_margin=0.05, **kw):
        self.robot, self.field, self.sigma, self.cutoff = robot, field, sigma_coll, cutoff_margin

    def factors(self, trajs):
        """the individual hinge factors [B, N, n_factors] whose sum is the cost (one per link sphere for an objects field, one per
        sphere and workspace FACE for the boundary field, one per sphere pair for the self-collision field): the residual vector of
        GPMP2's obstacle factors (oracle/gpmp.py)."""
        q = trajs[..., : self.robot.q_dim]
        pts = self.robot.link_points(q)  # [B, N, K, dim]
        radii = self.robot.radii.to(trajs.dtype)
        f = self.field
        if f.kind == "objects":
            return torch.relu(radii + self.cutoff - f.sdf(pts))
        if f.kind == "workspace":
            lo = pts - f.ws_min.to(trajs.dtype)
            hi = f.ws_max.to(trajs.dtype) - pts
            m = (radii + self.cutoff).unsqueeze(-1)
            return torch.cat([torch.relu(m - lo), torch.relu(m - hi)], dim=-1).flatten(-2)
        if f.kind == "self":
            a, b = pts[..., f.pairs[:, 0], :], pts[..., f.pairs[:, 1], :]
            return torch.relu(radii[f.pairs[:, 0]] + radii[f.pairs[:, 1]] - torch.linalg.norm(a - b, dim=-1))
        raise NotImplementedError(f.kind)

    def __call__(self, trajs):
        return self.factors(trajs).sum((-1, -2)) / (self.sigma ** 2)


class CostGPTrajectory:
    """half_factor: GPMP2 writes the prior as 1/2 sum e^T Qinv e; whether mp_baselines keeps the 1/2 is undecidable from the
    reference tree (empty submodule), so it is a switch (default False), mirrored by mpdx_guide_params.gp_half_factor."""

    def __init__(self, robot, n_support_points, dt, sigma_gp=1.0, half_factor=False, **kw):
        self.robot, self.dt, self.sigma = robot, float(dt), sigma_gp
        self.half_factor = bool(half_factor)

    def __call__(self, trajs):
        qd, dt = self.robot.q_dim, self.dt
        q, v = trajs[..., :qd], trajs[..., qd:]
        eq = q[:, 1:] - q[:, :-1] - dt * v[:, :-1]
        ev = v[:, 1:] - v[:, :-1]
        c = (12.0 / dt ** 3) * (eq * eq).sum(-1) - (12.0 / dt ** 2) * (eq * ev).sum(-1) + (4.0 / dt) * (ev * ev).sum(-1)
        return (0.5 if self.half_factor else 1.0) * c.sum(-1) / (self.sigma ** 2)


class CostComposite:
    """Collision terms see the interpolated trajectory, every other term the support points (call-site contract
    of guides.py:182-190)."""

    def __init__(self, robot, n_support_points, cost_list, weights_cost_l=None, **kw):
        self.cost_l, self.weight_l = list(cost_list), list(weights_cost_l)

    def __call__(self, trajs, x_interpolated=None, return_invidual_costs_and_weights=False, **kw):
        out = []
        for c in self.cost_l:
            src = x_interpolated if (isinstance(c, CostCollision) and x_interpolated is not None) else trajs
            out.append(c(src))
        if return_invidual_costs_and_weights:
            return out, self.weight_l
        return sum(w * c for w, c in zip(self.weight_l, out))
