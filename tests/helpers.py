"""Shared test helpers: synthetic weights, toy costs (same formulas as tests/golden/make_golden.py)."""
import numpy as np
import torch

from mpd_public_amd import synthetic as syn
from oracle.unet import unet_param_shapes

DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}


def synth_sd(D, opt):
    return syn.synth_state_dict(unet_param_shapes(D, 32, DIM_MULTS[opt]))


def toy_cost(x, x_interpolated=None, return_invidual_costs_and_weights=False, **kw):
    q = x.shape[-1] // 2
    c1 = (x_interpolated[..., :q] - 0.3).pow(2).sum((-1, -2)) * 3.0
    c2 = (x[:, 1:, :] - x[:, :-1, :]).pow(2).sum((-1, -2)) * 0.5 + (x[..., q:]).abs().sum((-1, -2)) * 0.01
    return [c1, c2], [1e-2, 3e-3]


def t(name, shape, kind="normal", scale=1.0):
    return torch.from_numpy(syn.synth_tensor(name, shape, kind, scale))


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


# ---------------------------------------------------------------------------------------------- guidance helpers
def oracle_guide(dataset, w_coll=1e-2, w_smooth=1e-7, clip_grad=True, interpolate=True, n_interp=128, dtype=torch.float32,
                 clip_grad_rule="norm", max_grad_value=0.1, gp_half_factor=False):
    """The oracle's guide (autograd over oracle/costs.py) for the same task as a product `TrajectoryDataset`."""
    from oracle import costs as oc
    from oracle.guide import GuideManager
    from oracle.normalizer import LimitsNormalizer
    from mpd_public_amd import _lib
    env, rob = dataset.env, dataset.robot
    dim = env.dim
    robot = oc.RobotPanda() if rob.name == "RobotPanda" else oc.RobotPointMass(rob.q_dim, rob.link_margin)
    robot.radii = robot.radii.to(dtype)
    cl, wl = [], []
    for f in dataset.task.get_collision_fields():
        if f.kind == _lib.FIELD_OBJECTS:
            o = f.objects
            fld = oc.ObjectField(torch.tensor(o.sphere_centers[:, :dim], dtype=dtype), torch.tensor(o.sphere_radii, dtype=dtype),
                                 torch.tensor(o.box_centers[:, :dim], dtype=dtype), torch.tensor(o.box_half[:, :dim], dtype=dtype))
        elif f.kind == _lib.FIELD_WORKSPACE:
            fld = oc.WorkspaceField(torch.tensor(f.ws_min, dtype=dtype), torch.tensor(f.ws_max, dtype=dtype))
        else:
            fld = oc.SelfField(torch.tensor(oc.PANDA_SELF_PAIRS, dtype=torch.long))
        cl.append(oc.CostCollision(robot, 64, field=fld, sigma_coll=1.0, cutoff_margin=dataset.task.obstacle_cutoff_margin))
        wl.append(w_coll)
    dt = 5.0 / dataset.n_support_points
    cl.append(oc.CostGPTrajectory(robot, 64, dt, sigma_gp=1.0, half_factor=gp_half_factor))
    wl.append(w_smooth)
    comp = oc.CostComposite(robot, 64, cl, weights_cost_l=wl)
    nrm = LimitsNormalizer(dataset.normalizer.mins.cpu(), dataset.normalizer.maxs.cpu())
    nrm.mins, nrm.maxs = nrm.mins.to(dtype), nrm.maxs.to(dtype)
    return GuideManager(nrm, comp, clip_grad=clip_grad, interpolate=interpolate, n_interp=n_interp, clip_grad_rule=clip_grad_rule,
                        max_grad_value=max_grad_value), comp


def product_guide(dataset, w_coll=1e-2, w_smooth=1e-7, clip_grad=True, interpolate=True, clip_grad_rule="norm", max_grad_value=0.1,
                  gp_half_factor=False):
    """The product guide built exactly as scripts/inference/inference.py:188-236 builds it."""
    import mpd_public_amd as m
    H = dataset.n_support_points
    dt = 5.0 / H
    costs = [m.CostCollision(dataset.robot, H, field=f, sigma_coll=1.0) for f in dataset.task.get_collision_fields()]
    weights = [w_coll] * len(costs)
    costs.append(m.CostGPTrajectory(dataset.robot, H, dt, sigma_gp=1.0, half_factor=gp_half_factor))
    weights.append(w_smooth)
    comp = m.CostComposite(dataset.robot, H, costs, weights_cost_l=weights)
    from math import ceil
    return m.GuideManagerTrajectoriesWithVelocity(dataset, comp, clip_grad=clip_grad, clip_grad_rule=clip_grad_rule, max_grad_value=max_grad_value,
                                                  interpolate_trajectories_for_collision=interpolate,
                                                  num_interpolated_points=ceil(H * 1.5))  # misspelt kwarg, as inference.py:234


def obstacle_hugging_trajs(dataset, B, seed="traj", scale=1.0):
    """Normalised [B,64,D] trajectories: straight lines between random configurations + noise, so that many waypoints
    sit inside obstacle margins / outside the workspace (every hinge branch is exercised)."""
    D = dataset.state_dim
    qd = D // 2
    from oracle.normalizer import LimitsNormalizer as _N
    cpu_norm = _N(dataset.normalizer.mins.cpu(), dataset.normalizer.maxs.cpu())
    a = t(f"{seed}/a", (B, 1, qd), "uniform", 0.95)
    b = t(f"{seed}/b", (B, 1, qd), "uniform", 0.95)
    if dataset.robot.name == "RobotPointMass":  # trajectory 0: from outside the workspace corner into an extra object
        a[0, 0, :] = -1.1
        c = torch.tensor(dataset.env.obj_extra.sphere_centers[0, :qd])
        b[0, 0, :] = cpu_norm.normalize(torch.cat([c, c]))[:qd]
    if dataset.robot.name == "RobotPanda":  # trajectory 0: from a self-colliding pose to a pose that leaves the workspace
        qa, qb = panda_probe_configs(dataset)
        nq = cpu_norm.normalize(torch.stack([torch.cat([qa, qa]), torch.cat([qb, qb])]))[:, :qd]
        a[0, 0, :], b[0, 0, :] = nq[0], nq[1]
    s = torch.linspace(0, 1, 64).reshape(1, 64, 1)
    pos = a + (b - a) * s + 0.05 * t(f"{seed}/n", (B, 64, qd))
    vel = 0.3 * t(f"{seed}/v", (B, 64, qd))
    return (scale * torch.cat([pos, vel], -1)).contiguous()


_PROBE = {}


def panda_probe_configs(dataset):
    """(q with an active self-collision hinge, q with an active workspace-boundary hinge), found by a deterministic
    scan of hash-uniform joint samples through the oracle's FK."""
    if "panda" in _PROBE:
        return _PROBE["panda"]
    from oracle import costs as oc
    rob = oc.RobotPanda()
    lo, hi = dataset.normalizer.mins[:7].cpu(), dataset.normalizer.maxs[:7].cpu()
    q = lo + (hi - lo) * (t("panda_probe", (4000, 7), "uniform") * 0.5 + 0.5)
    P = rob.link_points(q)
    pairs = torch.tensor(oc.PANDA_SELF_PAIRS)
    d = torch.linalg.norm(P[:, pairs[:, 0]] - P[:, pairs[:, 1]], dim=-1)
    selfc = torch.relu(rob.radii[pairs[:, 0]] + rob.radii[pairs[:, 1]] - d).sum(-1)
    m = (rob.radii + dataset.task.obstacle_cutoff_margin).unsqueeze(-1)
    wsc = (torch.relu(m - (P - torch.tensor(dataset.task.ws_min))) + torch.relu(m - (torch.tensor(dataset.task.ws_max) - P))).sum((-1, -2))
    assert selfc.max() > 0 and wsc.max() > 0
    _PROBE["panda"] = (q[selfc.argmax()], q[wsc.argmax()])
    return _PROBE["panda"]


def oracle_plan_metrics(dataset, xu, n_check=256):
    """The oracle's restatement of the post-loop metrics (inference.py:288-297,311-316) on UNNORMALISED trajectories xu [B,H,D]
    (CPU tensor): (#colliding interpolated waypoints [B], path length [B], smoothness [B]).  A waypoint collides iff some hinge
    with margin = link radius (no cutoff margin) is active on the n_check-point interpolation."""
    from oracle import costs as oc
    from oracle.guide import interpolate_points_v1
    _, comp = oracle_guide(dataset, dtype=xu.dtype)
    qd = dataset.state_dim // 2
    q, v = xu[..., :qd], xu[..., qd:]
    plen = torch.linalg.norm(q[:, 1:] - q[:, :-1], dim=-1).sum(-1)
    smooth = torch.linalg.norm(v[:, 1:] - v[:, :-1], dim=-1).sum(-1)
    xi = interpolate_points_v1(xu, n_check)
    hit = torch.zeros(xi.shape[:2], dtype=torch.bool)
    for term in comp.cost_l:
        if isinstance(term, oc.CostCollision):
            term.cutoff = 0.0
            per_point = torch.stack([term(xi[:, i:i + 1]) for i in range(xi.shape[1])], 1)
            hit |= per_point > 0
    return hit.sum(1), plen, smooth


import contextlib
import os


@contextlib.contextmanager
def kernel_path(fused: bool):
    """Run the enclosed product calls on the fused level programs (default) or on the per-layer conv kernels everywhere
    (MPDX_FUSED=0; libmpdx reads the variable on every pass)."""
    old = os.environ.get("MPDX_FUSED")
    os.environ["MPDX_FUSED"] = "1" if fused else "0"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("MPDX_FUSED", None)
        else:
            os.environ["MPDX_FUSED"] = old
