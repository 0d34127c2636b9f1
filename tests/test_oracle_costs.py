"""Analytic validation of oracle/costs.py (the restated, PARITY-UNPINNED half): closed forms, finite differences,
dense-matrix GP form, FK invariants.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import costs as oc
from oracle.guide import interpolate_points_v1
from helpers import oracle_guide, obstacle_hugging_trajs, t


def test_sdf_closed_forms():
    p = torch.tensor([[0.5, 0.0], [0.0, 0.0], [2.0, 2.0]])
    s = oc.sdf_spheres(p, torch.tensor([[0.0, 0.0]]), torch.tensor([0.2]))
    np.testing.assert_allclose(s[:, 0].numpy(), [0.3, -0.2, math.sqrt(8) - 0.2], rtol=1e-6)
    b = oc.sdf_boxes(p, torch.tensor([[0.0, 0.0]]), torch.tensor([[0.25, 0.1]]))
    # (0.5,0): outside along x by 0.25; (0,0): inside, nearest face at 0.1; (2,2): corner distance
    np.testing.assert_allclose(b[:, 0].numpy(), [0.25, -0.1, math.hypot(1.75, 1.9)], rtol=1e-6)


def test_interpolation_endpoints_and_midpoints():
    x = torch.arange(64, dtype=torch.float32).reshape(1, 64, 1)
    y = interpolate_points_v1(x, 128)[0, :, 0]
    assert y[0] == 0 and y[-1] == 63
    np.testing.assert_allclose(y.numpy(), np.arange(128) * 63 / 127, rtol=1e-6, atol=1e-5)


def test_gp_cost_equals_dense_quadratic_form():
    """c = sum_i e_i^T Qinv e_i with the GPMP2 constant-velocity blocks, against an explicit dense construction."""
    qd, H, dt = 2, 64, 5.0 / 64
    x = t("gp_x", (3, H, 2 * qd)).double()
    c = oc.CostGPTrajectory(oc.RobotPointMass(qd), H, dt)(x)
    I = np.eye(qd)
    Phi = np.block([[I, dt * I], [np.zeros((qd, qd)), I]])
    Qinv = np.block([[12 / dt ** 3 * I, -6 / dt ** 2 * I], [-6 / dt ** 2 * I, 4 / dt * I]])
    ref = []
    for b in range(3):
        xb = x[b].numpy()
        tot = 0.0
        for i in range(H - 1):
            e = xb[i + 1] - Phi @ xb[i]
            tot += e @ Qinv @ e
        ref.append(tot)
    np.testing.assert_allclose(c.numpy(), ref, rtol=1e-10)


def test_panda_fk_invariants():
    rob = oc.RobotPanda()
    q = t("fk_q", (5, 7), "uniform", 2.0).double()
    fr = rob.frames(q)
    O = [f[..., :3, 3] for f in fr]
    # link lengths of the kinematic chain are joint-independent (d3, hypot(a4, d5) ...)
    np.testing.assert_allclose(torch.linalg.norm(O[2] - O[1], dim=-1).numpy(), 0.316, rtol=1e-9)
    np.testing.assert_allclose(torch.linalg.norm(O[4] - O[3], dim=-1).numpy(), math.hypot(0.0825, 0.384), rtol=1e-9)
    np.testing.assert_allclose(torch.linalg.norm(O[6] - O[5], dim=-1).numpy(), 0.088, rtol=1e-9)
    for f in fr:  # proper rotations
        R = f[..., :3, :3]
        np.testing.assert_allclose((R @ R.transpose(-1, -2)).numpy(), np.broadcast_to(np.eye(3), R.shape), atol=1e-12)
    # q = 0: arm straight up, wrist offset 0.088 along x
    z = rob.frames(torch.zeros(1, 7, dtype=torch.float64))
    np.testing.assert_allclose(z[6][0, :3, 3].numpy(), [0.088, 0.0, 0.333 + 0.316 + 0.384], atol=1e-12)


@pytest.mark.parametrize("env_id,robot_id", [("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
def test_cost_gradients_match_finite_differences(env_id, robot_id):
    """autograd of every restated cost term vs central differences (fp64), on trajectories that touch obstacles."""
    import mpd_public_amd as m
    ds = m.TrajectoryDataset(env_id, robot_id)
    gm, comp = oracle_guide(ds, dtype=torch.float64)
    xn = obstacle_hugging_trajs(ds, 2, seed=f"fd/{robot_id}", scale=1.0).double()
    x = gm.normalizer.unnormalize(xn)
    active = 0
    for term in comp.cost_l:
        def f(z):
            zi = interpolate_points_v1(z, 128) if isinstance(term, oc.CostCollision) else z
            return term(zi).sum()
        xg = x.clone().requires_grad_(True)
        g = torch.autograd.grad(f(xg), xg)[0]
        # probe a handful of coordinates with the largest gradient + a few random ones
        flat = g.abs().flatten()
        idx = torch.cat([flat.topk(6).indices, torch.randint(0, flat.numel(), (6,), generator=torch.Generator().manual_seed(0))])
        for k in idx.tolist():
            e = torch.zeros_like(x).flatten()
            h = 1e-6
            e[k] = h
            e = e.reshape(x.shape)
            fd = (f(x + e) - f(x - e)) / (2 * h)
            assert abs(fd - g.flatten()[k]) <= 1e-3 * max(1.0, abs(fd)), (type(term).__name__, k, float(fd), float(g.flatten()[k]))
        active += int(flat.max() > 0)
    assert active == len(comp.cost_l), "every cost term must be active on the probe trajectories"


def test_oracle_metrics_closed_forms():
    """oracle/metrics.py (un-vendored torch_robotics metrics, inference.py:24,311-327) against closed forms."""
    from oracle import metrics as om
    H, qd = 16, 3
    s = np.linspace(0.0, 1.0, H)[:, None]
    line = np.concatenate([s * np.array([[3.0, 4.0, 0.0]]), np.zeros((H, qd))], -1)            # straight line of length 5, zero velocity
    c = np.array([0.3, -0.4, 1.2])
    two = np.stack([line, line + np.concatenate([c, np.zeros(qd)])[None]])                      # a second copy offset by c
    np.testing.assert_allclose(om.compute_path_length(two, qd), [5.0, 5.0], rtol=1e-12)
    np.testing.assert_allclose(om.compute_smoothness(two, qd), [0.0, 0.0], atol=1e-15)
    # two samples a, a + c: unbiased variance per coordinate c_j^2 / 2 -> H * |c|^2 / 2
    np.testing.assert_allclose(om.compute_variance_waypoints(two, qd), H * (c ** 2).sum() / 2, rtol=1e-12)
    assert om.compute_variance_waypoints(two[:1], qd) == 0.0
    # three copies at offsets 0, c, 3c: pair distances |c|, 3|c|, 2|c| at every waypoint -> variance |c|^2 per waypoint
    three = np.stack([line, line + np.concatenate([c, np.zeros(qd)])[None], line + np.concatenate([3 * c, np.zeros(qd)])[None]])
    np.testing.assert_allclose(om.compute_variance_waypoints(three, qd, "pairwise_distance"), H * (c ** 2).sum(), rtol=1e-12)
    assert om.compute_variance_waypoints(np.stack([line, line, line]), qd, "pairwise_distance") == 0.0
