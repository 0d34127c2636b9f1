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
    assert 1 <= used <= 6000 and int(rrt.count.min()) >= 1 and int(rrt.count.max()) <= rrt.M
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


def test_generation_experiment_entry_writes_metadata(tmp_path):
    """experiment(...) of scripts/generate_data/generate_trajectories.py:170-246: metadata.yaml beside the trajectory files, unknown keyword arguments swallowed."""
    import yaml
    from mpd_public_amd.generate_trajectories import experiment
    n_coll, n_free = experiment(env_id="EnvSimple2D", robot_id="RobotPointMass", num_trajectories=8, threshold_start_goal_pos=1.0, results_dir=str(tmp_path),
                                seed=5, debug=False, gpmp_opt_iters=150, some_launcher_key=1)
    md = yaml.safe_load(open(tmp_path / "metadata.yaml"))
    assert md["env_id"] == "EnvSimple2D" and md["num_trajectories"] == 8 and md["num_trajectories_generated"] == n_coll + n_free == 8
    assert md["num_trajectories_generated_free"] == n_free and (tmp_path / "trajs-free.pt").exists() and (tmp_path / "results_data_dict.pickle").exists()


def test_rrt_connect_full_tree_or_iteration_cap_reports_unsolved():
    """a problem whose tree fills up (or that runs out of iterations) must END unsolved - never record a link through a
    node that was not inserted.  A tiny node budget in the narrow-passage environment forces both outcomes."""
    from mpd_public_amd.generate_trajectories import RRTConnectBatch, edges_free
    ds = _dataset("EnvNarrowPassageDense2D", "RobotPointMass")
    start, goal = _start_goal(ds, 3)
    rrt = RRTConnectBatch(ds.task, start, goal, 32, step_size=0.05, max_nodes=24, generator=torch.Generator(device="cuda").manual_seed(1))
    rrt.grow(max_iters=400)
    cnt, link, done = rrt.count.cpu(), rrt.link.cpu(), rrt.done.cpu()
    assert int(cnt.max()) <= 24
    assert bool(((link >= 0).all(1) == done).all()) and bool(((link < 0).all(1) == ~done).all())
    assert not bool(done.all()), "the budget was meant to be too small for some problems"
    for i, p in enumerate(rrt.paths()):
        if p is None:
            continue
        assert torch.equal(p[0], start.cpu()) and torch.equal(p[-1], goal.cpu())
        assert bool(edges_free(ds.task, p[:-1].cuda().contiguous(), p[1:].cuda().contiguous(), 64).all())


@pytest.mark.parametrize("env_id,robot_id,H,n_interp", [("EnvDense2D", "RobotPointMass", 64, 128), ("EnvSpheres3D", "RobotPanda", 16, 32),
                                                         # the linearisation's thread mappings: 4 N = 512 threads (above), N = 96 (parts not wave-aligned), N = 256 (two
                                                         # threads per point: 4 N > 512), the Panda at the generator's size (H = 64, 128 points: the four-part mapping)
                                                         ("EnvDense2D", "RobotPointMass", 64, 96), ("EnvDense2D", "RobotPointMass", 64, 256),
                                                         ("EnvSpheres3D", "RobotPanda", 64, 128)])
def test_gpmp2_lm_step_vs_oracle(env_id, robot_id, H, n_interp):
    """One Levenberg-Marquardt step of the HIP kernel (hand-derived factor Jacobians, block-tridiagonal system assembled and solved
    in LDS, fp32) == oracle/gpmp.py (forward-mode autograd Jacobian of the stacked residuals, dense float64 solve), and the
    objective the kernel reports == 1/2 |r|^2 of the oracle."""
    import ctypes as C
    from mpd_public_amd import _lib
    from mpd_public_amd.generate_trajectories import GPMP2
    from helpers import obstacle_hugging_trajs
    from oracle import gpmp as ogpmp
    from oracle.normalizer import LimitsNormalizer
    ds = _dataset(env_id, robot_id)
    B, dt = 3, 5.0 / 64
    xn = obstacle_hugging_trajs(ds, B, seed=f"gpmp2/{env_id}", scale=0.9)
    xu = LimitsNormalizer(ds.normalizer.mins.cpu(), ds.normalizer.maxs.cpu()).unnormalize(xn)      # raw robot units
    xu = xu[:, ::64 // H].contiguous()
    sigma_gp, sigma_obs, lam = 1.0, 2e-2, 1e-2
    ds.n_support_points = H
    opt = GPMP2(ds, dt, sigma_gp=sigma_gp, sigma_obs=sigma_obs, n_interp=n_interp, lambda_init=lam, device="cuda")
    _, comp = oracle_guide(ds, 1.0, 1.0, clip_grad=False, dtype=torch.float64)
    coll = comp.cost_l[:-1]
    for c in coll:
        c.cutoff = ds.task.obstacle_cutoff_margin
    robot = coll[0].robot
    x = xu.cuda().contiguous().clone()
    delta = torch.zeros_like(x)
    state = torch.zeros((B, 4), device="cuda")
    state[:, 0], state[:, 1] = 3.0e38, lam
    _lib.check(_lib.load().mpdx_gpmp_step(C.byref(opt.gp), C.byref(opt.opts), x.data_ptr(), delta.data_ptr(), state.data_ptr(), B, H,
                                          ds.state_dim, 1, _lib.current_stream()), "mpdx_gpmp_step")
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), xu)                      # the first call accepts the (zero) proposal: the point is unchanged
    for b in range(B):
        want, F = ogpmp.lm_step(xu[b].double(), robot, coll, dt, sigma_gp, sigma_obs, n_interp, lam)
        got = delta[b].cpu().double()
        assert float(F) > 0 and abs(float(state[b, 0]) - float(F)) <= 2e-4 * float(F), (b, float(state[b, 0]), float(F))
        assert not got[0].any() and not got[-1].any()
        scale = float(want.abs().max())
        assert scale > 1e-4
        err = float((got - want).abs().max())
        assert err <= 2e-2 * scale, (b, err, scale)


@pytest.mark.parametrize("env_id,robot_id", [("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
def test_gpmp2_descends_monotonically_and_repairs_collisions(env_id, robot_id):
    """LM accepts only steps that lower F (the kernel's own objective: checked against the oracle's at the end), keeps the end
    states fixed, and drives RRT-initialised trajectories to collision-free ones."""
    from mpd_public_amd.generate_trajectories import GPMP2, RRTConnectBatch, shortcut_path, resample_path
    from oracle import gpmp as ogpmp
    ds = _dataset(env_id, robot_id)
    start, goal = _start_goal(ds, 3)   # (seed 7 draws a Panda pair that 6000 iterations do not connect: RRT may fail, that is not the subject here)
    n, dt = 12, 5.0 / 64
    rrt = RRTConnectBatch(ds.task, start, goal, n, step_size=0.1 if ds.robot.q_dim <= 3 else 0.25, generator=torch.Generator(device="cuda").manual_seed(2))
    rrt.grow(max_iters=6000)
    assert bool(rrt.done.all())
    x0 = torch.stack([resample_path(shortcut_path(ds.task, p), 64, dt) for p in rrt.paths()]).cuda()
    opt = GPMP2(ds, dt, device="cuda")
    Fs = []
    x = x0
    for k in range(6):
        x = opt.optimize(x, opt_iters=40 if k else 1)
        Fs.append(opt.state[:, 0].clone())
        # (each optimize() call restarts lambda; F of the accepted point can only go down within and across calls)
    Fs = torch.stack(Fs).cpu()
    assert bool((Fs[1:] <= Fs[:-1] * (1 + 1e-5)).all()), Fs
    assert float((Fs[-1] / Fs[0]).max()) < 0.9
    assert torch.equal(x[:, 0], x0[:, 0]) and torch.equal(x[:, -1], x0[:, -1])
    # a local method on 128 interpolated points, judged on 256: a resampled path that cuts a corner may stay in a stiff local minimum and
    # a smoothed one may graze an obstacle between two factor points (Panda, this context: 10-11 of 12 free before and after); the entry
    # classifies the outcome (trajs-free / trajs-collision), as the reference's does
    f0, f1 = ds.task.compute_fraction_free_trajs(x0), ds.task.compute_fraction_free_trajs(x)
    assert f1 >= 0.7 and f1 >= f0 - 0.15, (f0, f1)
    # the objective the kernel tracks is the oracle's
    _, comp = oracle_guide(ds, 1.0, 1.0, clip_grad=False, dtype=torch.float64)
    coll = comp.cost_l[:-1]
    for c in coll:
        c.cutoff = ds.task.obstacle_cutoff_margin
    F0 = float(ogpmp.objective(x[0].cpu().double(), coll[0].robot, coll, dt, 1.0, opt.opts.sigma_obs, 128))
    assert abs(float(Fs[-1, 0]) - F0) <= 1e-3 * F0 + 1e-6, (float(Fs[-1, 0]), F0)


@pytest.mark.parametrize("env_id,robot_id", [("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
def test_rrt_paths_on_device_match_the_host_algorithms_and_are_collision_free(env_id, robot_id):
    """mpdx_rrt_paths (path extraction + greedy shortcutting + arc-length resampling in ONE launch) against the host restatements
    shortcut_path / resample_path on the same trees: same shortcut nodes, same trajectories to fp32 rounding; every segment between
    consecutive shortcut nodes is re-checked 4x finer; an unsolved problem becomes the straight line."""
    import mpd_public_amd as m
    from mpd_public_amd.generate_trajectories import RRTConnectBatch, edges_free, shortcut_path, resample_path
    ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
    task = ds.task
    gen = torch.Generator(device="cuda").manual_seed(11)
    q = task.random_coll_free_q(n_samples=2, device="cuda", generator=gen)
    n, H, dt = 12, 64, 5.0 / 64
    rrt = RRTConnectBatch(task, q[0], q[1], n, step_size=0.1 if ds.robot.q_dim <= 3 else 0.25, generator=gen)
    rrt.grow(max_iters=6000)
    assert int(rrt.done.sum()) >= n - 2
    rrt.link[0] = -1; rrt.done[0] = False        # an "unsolved" problem
    tr, plen = rrt.trajectories(H, dt, return_path_len=True)
    tr, plen = tr.cpu(), plen.cpu()
    assert tr.shape == (n, H, 2 * ds.robot.q_dim) and bool(torch.isfinite(tr).all())
    qd = ds.robot.q_dim
    line = torch.stack([q[0].cpu(), q[1].cpu()])
    assert int(plen[0]) == 2
    np.testing.assert_allclose(tr[0].numpy(), resample_path(line, H, dt).numpy(), rtol=0, atol=2e-5)
    same = 0
    for i, p in enumerate(rrt.paths()):
        if p is None:
            continue
        sc = shortcut_path(task, p)
        assert torch.equal(tr[i, 0, :qd], q[0].cpu()) and torch.equal(tr[i, -1, :qd], q[1].cpu())
        assert not tr[i, 0, qd:].any() and not tr[i, -1, qd:].any()
        if int(plen[i]) == sc.shape[0]:   # same greedy decisions (an edge within fp32 rounding of an obstacle may be judged differently)
            same += 1
            np.testing.assert_allclose(tr[i].numpy(), resample_path(sc, H, dt).numpy(), rtol=0, atol=5e-4)
        # the device trajectory itself: its positions lie on collision-free segments (checked between consecutive support points, 8 checks each)
        pts = tr[i, :, :qd].cuda().contiguous()
        assert bool(edges_free(task, pts[:-1].contiguous(), pts[1:].contiguous(), n_edge_checks=8).all()), i
    assert same >= n - 3, same
