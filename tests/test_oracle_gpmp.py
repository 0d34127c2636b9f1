"""CPU checks of oracle/gpmp.py (the GPMP2 restatement the HIP planner kernel is compared with; PARITY UNPINNED - the reference's GPMP2 is
un-vendored - so the oracle is validated by the algorithm's own identities): the prior part is quadratic, so one undamped Gauss-Newton
step lands on its minimiser; the stacked residuals' forward-mode Jacobian equals finite differences; a damped step lowers the objective."""
import numpy as np
import torch

from oracle import costs as oc
from oracle import gpmp as og

DT = 5.0 / 64


def _field_2d():
    return oc.ObjectField(torch.tensor([[0.1, 0.0], [-0.4, 0.3]], dtype=torch.float64), torch.tensor([0.2, 0.15], dtype=torch.float64),
                          torch.tensor([[0.5, -0.4]], dtype=torch.float64), torch.tensor([[0.1, 0.2]], dtype=torch.float64))


def _traj(H=16, seed=0, through_obstacle=True):
    g = torch.Generator().manual_seed(seed)
    a, b = torch.tensor([-0.8, -0.1], dtype=torch.float64), torch.tensor([0.8, 0.1], dtype=torch.float64)
    s = torch.linspace(0, 1, H, dtype=torch.float64)[:, None]
    pos = a + (b - a) * s + (0.0 if through_obstacle else 0.6) + 0.02 * torch.randn((H, 2), generator=g, dtype=torch.float64)
    vel = 0.2 * torch.randn((H, 2), generator=g, dtype=torch.float64)
    vel[0] = vel[-1] = 0.0
    return torch.cat([pos, vel], -1)


def test_prior_only_step_reaches_the_minimiser():
    """No collision factors: F is quadratic in the free states, so one Gauss-Newton step with lambda = 0 makes the gradient vanish, and the
    minimiser of the constant-velocity prior between fixed end states has (numerically) constant acceleration-free segments: its
    objective is below every perturbation's."""
    robot = oc.RobotPointMass(2, 0.01)
    th = _traj()
    delta, F0 = og.lm_step(th, robot, [], DT, 1.0, 1.0, 0, 0.0)
    th1 = th + delta
    assert not delta[0].any() and not delta[-1].any()
    free = th1[1:-1].clone().requires_grad_(True)
    F1 = og.objective(torch.cat([th1[:1], free, th1[-1:]]), robot, [], DT, 1.0, 1.0, 0)
    (g,) = torch.autograd.grad(F1, free)
    assert float(F1) < float(F0)
    assert float(g.abs().max()) < 1e-6 * max(1.0, float(F0))
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        pert = th1.clone()
        pert[1:-1] += 1e-3 * torch.randn(pert[1:-1].shape, generator=gen, dtype=torch.float64)
        assert float(og.objective(pert, robot, [], DT, 1.0, 1.0, 0)) > float(F1)


def test_residual_jacobian_equals_finite_differences_and_step_descends():
    robot = oc.RobotPointMass(2, 0.01)
    robot.radii = robot.radii.double()
    coll = [oc.CostCollision(robot, 16, field=_field_2d(), cutoff_margin=0.05),
            oc.CostCollision(robot, 16, field=oc.WorkspaceField(torch.tensor([-1.0, -1.0], dtype=torch.float64), torch.tensor([0.85, 1.0], dtype=torch.float64)),
                             cutoff_margin=0.05)]
    th = _traj()
    H, D = th.shape

    def r_of(free):
        return og.residuals(torch.cat([th[:1], free.reshape(H - 2, D), th[-1:]]), robot, coll, DT, 1.0, 0.05, 32)
    x0 = th[1:-1].reshape(-1).clone()
    J = torch.func.jacfwd(r_of)(x0)
    r0 = r_of(x0)
    assert float((r0[2 * (H - 1) * 2:] > 0).sum()) > 3, "some collision factors must be active"
    eps = 1e-6
    gen = torch.Generator().manual_seed(2)
    for _ in range(6):
        v = torch.randn(x0.shape, generator=gen, dtype=torch.float64)
        fd = (r_of(x0 + eps * v) - r_of(x0 - eps * v)) / (2 * eps)
        # hinge kinks: compare only where the factor's activity does not change within the probe
        same = (r_of(x0 + eps * v) > 0) == (r_of(x0 - eps * v) > 0)
        np.testing.assert_allclose((J @ v)[same].numpy(), fd[same].numpy(), rtol=1e-5, atol=1e-6)
    d, F0 = og.lm_step(th, robot, coll, DT, 1.0, 0.05, 32, 1e-2)
    F1 = og.objective(th + d, robot, coll, DT, 1.0, 0.05, 32)
    assert float(F1) < float(F0)
    assert abs(float(F0) - 0.5 * float((r0 * r0).sum())) < 1e-9 * float(F0)
