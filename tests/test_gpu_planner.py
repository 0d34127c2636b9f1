"""GPU tests of the baseline planners behind generate_collision_free_trajectories: batched RRT-Connect with
the HIP collision checker, the GPMP-objective optimiser (HIP guide kernel in raw units) against the oracle's costs, and the entry.
The reference's planners are un-vendored (mp_baselines): PARITY UNPINNED - the checks are the algorithm's own invariants plus the
oracle's restatement of the objective."""
import pickle

import numpy as np
import pytest
import torch

from helpers import oracle_guide, t

pytestmark = pytest.mark.gpu


def _dataset(env_id, robot_id):
    import mpd_public_amd as m
    return m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})


def _start_goal(ds, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(200):
        q = ds.task.random_coll_free_q(n_samples=2, device="cuda", generator=g)
        if torch.linalg.norm(q[0] - q[1]) > ds.threshold_start_goal_pos:
            return q[0], q[1]
    raise AssertionError("no start/goal pair")


@pytest.mark.parametrize("env_id,robot_id,n", [("EnvDense2D", "RobotPointMass", 24), ("EnvSpheres3D", "RobotPanda", 8)])
def test_rrt_connect_batch_paths_are_valid(env_id, robot_id, n):
    from mpd_public_amd.generate_trajectories import RRTConnectBatch, edges_free, shortcut_path, resample_path
    ds = _dataset(env_id, robot_id)
    start, goal = _start_goal(ds, 11)
    step = 0.1 if ds.robot.q_dim <= 3 else 0.25
    rrt = RRTConnectBatch(ds.task, start, goal, n, step_size=step, generator=torch.Generator(device="cuda").manual_seed(5))
    used = rrt.grow(max_iters=6000)
    assert int(rrt.done.sum()) == n, f"{int(rrt.done.sum())}/{n} solved in {used} iterations"
    paths = rrt.paths()
    lens = set()
    for p in paths:
        assert p is not None and p.shape[1] == ds.robot.q_dim
        assert torch.equal(p[0], start.cpu()) and torch.equal(p[-1], goal.cpu())            # exact end points
        seg = torch.linalg.norm(p[1:] - p[:-1], dim=-1)
        assert float(seg.max()) <= step * (1 + 1e-4)                                          # every edge is one steer
        assert bool(edges_free(ds.task, p[:-1].cuda().contiguous(), p[1:].cuda().contiguous(), 64).all())   # re-checked 4x finer
        sc = shortcut_path(ds.task, p)
        assert sc.shape[0] <= p.shape[0] and torch.equal(sc[0], p[0]) and torch.equal(sc[-1], p[-1])
        assert bool(edges_free(ds.task, sc[:-1].cuda().contiguous(), sc[1:].cuda().contiguous(), 32).all())
        tr = resample_path(sc, 64, 5.0 / 64)
        assert tr.shape == (64, 2 * ds.robot.q_dim) and torch.equal(tr[0, :ds.robot.q_dim], p[0]) and torch.equal(tr[-1, :ds.robot.q_dim], p[-1])
        lens.add(p.shape[0])
    assert len(lens) > 1, "the n samples of a context are different random trees"


@pytest.mark.parametrize("env_id,robot_id", [("EnvSimple2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
def test_gpmp_optimizer_step_and_descent_vs_oracle(env_id, robot_id):
    """One optimiser iteration == x + oracle guide increment in RAW units (identity normaliser, un-clipped gradients, the optimiser's
    step sizes), and `opt_iters` iterations decrease the oracle's GPMP objective with end points fixed."""
    from mpd_public_amd.generate_trajectories import GPMPOptimizer
    from helpers import obstacle_hugging_trajs
    ds = _dataset(env_id, robot_id)
    H, dt = 64, 5.0 / 64
    xn = obstacle_hugging_trajs(ds, 6, seed=f"gpmp/{env_id}", scale=0.9)
    from oracle.normalizer import LimitsNormalizer
    xu = LimitsNormalizer(ds.normalizer.mins.cpu(), ds.normalizer.maxs.cpu()).unnormalize(xn)      # raw robot units
    opt = GPMPOptimizer(ds, dt, device="cuda")
    # the oracle guide with an identity normaliser and the optimiser's step sizes as weights
    og, comp = oracle_guide(ds, opt.step_coll, opt.step_gp, clip_grad=False, dtype=torch.float64)

    class _Id:
        def unnormalize(self, x):
            return x
    og.normalizer = _Id()
    inc = og(xu.double())
    want = xu.double() + inc
    want[:, 0], want[:, -1] = xu[:, 0].double(), xu[:, -1].double()
    got = opt.optimize(xu.cuda(), opt_iters=1).cpu().double()
    bad = (np.abs(got.numpy() - want.numpy()) > 2e-6 + 1e-3 * np.abs(inc.numpy())).any(-1)
    assert bad.mean() < 0.01, f"{bad.sum()} of {bad.size} waypoints differ"
    # descent on the oracle's objective (collision terms on the 128-point interpolation, GP prior on the support points)
    from oracle.guide import interpolate_points_v1

    def objective(x):
        cl, wl = comp(x, x_interpolated=interpolate_points_v1(x, 128), return_invidual_costs_and_weights=True)
        return sum(w * c for c, w in zip(cl, wl))
    x500, iters = opt.optimize(xu.cuda(), opt_iters=300, return_iterations=True)
    c0, c1 = objective(xu.double()), objective(x500.cpu().double())
    assert bool((c1 < c0).all()), (c0, c1)
    assert float((c1 / c0).max()) < 0.9
    assert torch.equal(x500[:, 0].cpu(), xu[:, 0]) and torch.equal(x500[:, -1].cpu(), xu[:, -1])
    assert iters.shape == (301, 6, H, ds.state_dim)


def test_generate_collision_free_trajectories_entry(tmp_path):
    from mpd_public_amd.generate_trajectories import generate_collision_free_trajectories
    n = 16
    n_coll, n_free = generate_collision_free_trajectories("EnvSimple2D", "RobotPointMass", n, str(tmp_path), gpmp_opt_iters=200, seed=3)
    assert n_coll + n_free == n
    last = generate_collision_free_trajectories.last
    assert last["rrt_solved"] == n
    assert n_free >= n // 2, (n_free, last["fraction_free"], last["collision_intensity"])
    free = torch.load(tmp_path / "trajs-free.pt")
    coll = torch.load(tmp_path / "trajs-collision.pt")
    assert free.shape == (n_free, 64, 4) and (coll.numel() == 0 or coll.shape == (n_coll, 64, 4))
    with open(tmp_path / "results_data_dict.pickle", "rb") as f:
        d = pickle.load(f)
    assert d["n_support_points"] == 64 and abs(d["dt"] - 5.0 / 64) < 1e-12 and d["duration"] == 5.0
    assert d["trajs_iters_free"].shape == (1, n_free, 64, 4)
    # every saved collision-free trajectory starts and ends at the context's start / goal with zero velocity
    assert torch.equal(free[:, 0], free[0:1, 0].expand(n_free, -1)) and torch.equal(free[:, -1], free[0:1, -1].expand(n_free, -1))
    assert not free[:, 0, 2:].any() and not free[:, -1, 2:].any()
    # the optimiser made the trajectories smoother than the resampled RRT paths (GP prior): mean acceleration energy drops
    acc = lambda x: torch.diff(x[..., :2], n=2, dim=1).pow(2).sum((-1, -2)).mean()   # noqa: E731
    assert float(acc(last["trajs_iters"][-1])) < float(acc(last["trajs_init"]))
