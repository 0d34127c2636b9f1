"""generate_collision_free_trajectories() - drop-in for scripts/generate_data/generate_trajectories.py:20-169 (dataset generation
with the reference's baseline planners: RRT-Connect initialisation + GPMP2 optimisation = `HybridPlanner`, :66-120).

Same arguments, same artefacts (`trajs-free.pt`, `trajs-collision.pt`, `results_data_dict.pickle` with the same keys, :131-152),
same return value (number of colliding / collision-free trajectories).  It doubles as the like-for-like PLANNER BASELINE of the
diffusion sampler on the same box.

The reference's planners live in the un-vendored `mp_baselines` submodule (empty in /root/reference: PARITY UNPINNED, as for the
guide's costs - DESIGN.md section 5).  What is built here, MI355X-first:

  * RRT-Connect (Kuffner & LaValle 2000): the n trajectories of a context are n independent bidirectional searches, each run
    START TO FINISH by one workgroup of ONE kernel launch (`mpdx_rrt_connect`, csrc/planner.hpp): Philox sampling, nearest
    neighbour over the LDS-resident trees, steering, edge collision checks (the metrics kernel's FK / SDF functions on
    `n_edge_checks` interpolated configurations) and the greedy connect loop all happen on the device - no host round trip per
    extension.  The reference's `MultiSampleBasedPlanner` runs its n RRTs one after the other in Python (:85-90).
  * GPMP2 (Mukadam et al. 2018): Levenberg-Marquardt on  1/2 |GP prior factors|^2 / sigma_gp^2 + 1/2 |hinge collision factors|^2 /
    sigma_obs^2  with start and goal states fixed (`GPMP2`, `mpdx_gpmp_step`): per iteration and trajectory one workgroup linearises
    every collision factor of the 128 interpolated points (FK Jacobians), assembles the BLOCK-TRIDIAGONAL normal equations over the 62
    free support states as d x d blocks in LDS and solves them there (block cyclic reduction).  `oracle/gpmp.py` restates one step (autograd
    Jacobian + dense solve).
  * `GPMPOptimizer`: first-order descent on the guide's (un-squared hinge) objective with the HIP guide kernel in raw units - kept as
    the cheap smoother / as the raw-unit test vehicle of the guide kernel; the entry uses GPMP2.

Rendering (`PlanningVisualizer`, :155-167) is out of scope.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
import time
from typing import List, Optional, Tuple

import torch

from . import _lib
from .datasets import TrajectoryDataset
from .guides import build_device_params
from .planning import CostCollision, CostGPTrajectory


# ------------------------------------------------------------------------------------------------ collision checking of edges
def edges_free(task, qa: torch.Tensor, qb: torch.Tensor, n_edge_checks: int = 16) -> torch.Tensor:
    """qa, qb: [n, q] configurations on the GPU -> bool [n]: the straight segment qa -> qb is collision free (checked on
    n_edge_checks interpolated configurations with the metrics kernel, link margin only - as task.get_trajs_collision_and_free)."""
    z = torch.zeros_like(qa)
    traj = torch.stack([torch.cat([qa, z], -1), torch.cat([qb, z], -1)], dim=1).contiguous()   # [n, 2, 2q]
    return task.trajectory_metrics(traj, n_check=n_edge_checks)[:, 0] == 0


class RRTConnectBatch:
    """n independent RRT-Connect problems (one start/goal pair, n samples - or per-problem starts/goals): ONE launch of
    rrt_connect_kernel grows all of them to completion (or to max_iters / a full tree)."""

    _launches = 0

    def __init__(self, task, start: torch.Tensor, goal: torch.Tensor, n: int, step_size: float = 0.1, max_nodes: int = 2048,
                 n_edge_checks: int = 16, generator: Optional[torch.Generator] = None):
        dev = start.device
        if dev.type != "cuda":
            raise RuntimeError("RRTConnectBatch runs on the GPU (libmpdx RRT-Connect kernel); there is no CPU fallback")
        self.task, self.n, self.step, self.M, self.nchk, self.gen = task, n, float(step_size), int(max_nodes), n_edge_checks, generator
        q = start.shape[-1]
        self.q = q
        self.lo, self.hi = task.q_limits(dev)
        self.start = (start.reshape(1, q).expand(n, q) if start.dim() == 1 else start).to(torch.float32).contiguous()
        self.goal = (goal.reshape(1, q).expand(n, q) if goal.dim() == 1 else goal).to(torch.float32).contiguous()
        self.nodes = torch.zeros((n, 2, self.M, q), device=dev)          # tree 0 grows from the start, tree 1 from the goal
        self.parent = torch.full((n, 2, self.M), -1, dtype=torch.int32, device=dev)
        self.count = torch.ones((n, 2), dtype=torch.int32, device=dev)
        self.link = torch.full((n, 2), -1, dtype=torch.int32, device=dev)  # node indices (tree 0, tree 1) where the trees met
        self.iters = torch.zeros(n, dtype=torch.int32, device=dev)
        self.done = torch.zeros(n, dtype=torch.bool, device=dev)
        self._gp = task._params(dev)   # robot + collision fields as the metrics kernel sees them (the task keeps the primitive table alive)

    def grow(self, max_iters: int = 4000, max_connect_steps: int = 64) -> int:
        """Runs the search; returns the largest number of iterations any problem used.  One iteration = extend the active tree
        towards a random sample, then connect the other tree greedily towards the new node (alternating trees).
        The search's random stream is keyed by (the generator's initial seed, the number of searches launched in this process so far): a fixed sequence of
        calls in a fresh process reproduces its trajectories - what the reference's global torch RNG behind fix_random_seed gives - while two searches of one
        process never share a stream, also with equal generator seeds."""
        o = _lib.RrtOpts()
        for j in range(self.q):
            o.q_lo[j], o.q_hi[j] = float(self.lo[j]), float(self.hi[j])
        o.step, o.max_nodes, o.max_iters, o.max_connect_steps, o.n_edge_checks = self.step, self.M, int(max_iters), int(max_connect_steps), int(self.nchk)
        RRTConnectBatch._launches += 1
        o.seed = (int(self.gen.initial_seed()) if self.gen is not None else 0) * 1000003 + RRTConnectBatch._launches
        _lib.check(_lib.load().mpdx_rrt_connect(C.byref(self._gp), C.byref(o), self.start.data_ptr(), self.goal.data_ptr(), self.nodes.data_ptr(),
                                                self.parent.data_ptr(), self.count.data_ptr(), self.link.data_ptr(), self.iters.data_ptr(),
                                                self.n, _lib.current_stream()), "mpdx_rrt_connect")
        self.done = self.link[:, 0] >= 0
        return int(self.iters.max())

    def trajectories(self, n_support_points: int, dt: float, n_edge_checks: int = 32, rounds: int = 3, return_path_len: bool = False):
        """The searches' results as [n, H, 2q] state trajectories, entirely on the device (`mpdx_rrt_paths`, csrc/planner.hpp rrt_path_kernel):
        path extraction from the two trees, greedy shortcutting (`shortcut_path`'s algorithm) and arc-length resampling with
        central-difference velocities (`resample_path`'s), one workgroup per problem; an unsolved problem becomes the straight line.
        Round 3 ran these steps as host loops over GPU edge checks (~170 ms of the 200 ms of a 100-problem narrow-passage batch)."""
        H = int(n_support_points)
        out = torch.empty((self.n, H, 2 * self.q), dtype=torch.float32, device=self.nodes.device)
        plen = torch.zeros(self.n, dtype=torch.int32, device=self.nodes.device)
        _lib.check(_lib.load().mpdx_rrt_paths(C.byref(self._gp), self.start.data_ptr(), self.goal.data_ptr(), self.nodes.data_ptr(), self.parent.data_ptr(),
                                              self.link.data_ptr(), out.data_ptr(), plen.data_ptr(), self.n, self.M, H, float(dt), int(n_edge_checks),
                                              int(rounds), _lib.current_stream()), "mpdx_rrt_paths")
        return (out, plen) if return_path_len else out

    def paths(self) -> List[Optional[torch.Tensor]]:
        """Per problem: [n_nodes, q] configurations from start to goal (None if the trees did not meet)."""
        nodes, parent, link, done = self.nodes.cpu(), self.parent.cpu().tolist(), self.link.cpu().tolist(), self.done.cpu().tolist()
        out: List[Optional[torch.Tensor]] = []
        for i in range(self.n):
            if not done[i]:
                out.append(None)
                continue
            branch = []
            for tree in (0, 1):
                seq, k = [], link[i][tree]
                while k >= 0:
                    seq.append(k)
                    k = parent[i][tree][k]
                branch.append(nodes[i, tree, seq])
            out.append(torch.cat([branch[0].flip(0), branch[1]], dim=0))   # start ... meeting node | other tree's branch ... goal
        return out


def shortcut_path(task, path: torch.Tensor, n_edge_checks: int = 32, rounds: int = 3) -> torch.Tensor:
    """Greedy shortcutting: drop every intermediate node whose neighbours see each other (batched edge checks on the GPU)."""
    p = path
    for _ in range(rounds):
        if p.shape[0] <= 2:
            break
        keep = [0]
        i = 0
        while i < p.shape[0] - 1:
            cand = torch.arange(i + 1, p.shape[0])
            free = edges_free(task, p[i].cuda().expand(len(cand), -1).contiguous(), p[cand].cuda().contiguous(), n_edge_checks).cpu()
            j = int(cand[free][-1]) if bool(free.any()) else i + 1
            keep.append(j)
            i = j
        if len(keep) == p.shape[0]:
            break
        p = p[keep]
    return p


def resample_path(path: torch.Tensor, n_support_points: int, dt: float) -> torch.Tensor:
    """[m, q] waypoints -> [H, 2q] state trajectory: uniform in arc length, velocities by central differences, zero at both ends
    (the state layout the GP prior and the dataset use: positions then velocities)."""
    seg = torch.linalg.norm(path[1:] - path[:-1], dim=-1)
    s = torch.cat([torch.zeros(1), torch.cumsum(seg, 0)])
    total = float(s[-1])
    u = torch.linspace(0.0, total, n_support_points)
    idx = torch.searchsorted(s, u, right=True).clamp(1, len(s) - 1)
    w = ((u - s[idx - 1]) / (s[idx] - s[idx - 1]).clamp_min(1e-12)).clamp(0, 1)[:, None]
    pos = path[idx - 1] * (1 - w) + path[idx] * w
    pos[0], pos[-1] = path[0], path[-1]
    vel = torch.zeros_like(pos)
    vel[1:-1] = (pos[2:] - pos[:-2]) / (2 * dt)
    return torch.cat([pos, vel], dim=-1)


# ------------------------------------------------------------------------------------------------ optimiser on the GPMP2 objective
class GPMPOptimizer:
    """Gradient descent on  sum_fields w_coll * hinge-collision(interpolated trajectory) + w_gp * GP-prior(trajectory)  with the HIP
    guide kernel in raw units; one launch per iteration for the whole batch, start / goal states hard-conditioned."""

    def __init__(self, dataset: TrajectoryDataset, dt: float, sigma_gp: float = 1.0, step_coll: float = 3e-3, step_gp: Optional[float] = None,
                 n_interp: int = 128, clip_grad: bool = False, max_grad_norm: float = 1.0, device="cuda"):
        # step sizes of plain gradient descent: the hinge gradient of a field is a unit vector per active link sphere; the GP
        # prior's Hessian has lambda_max ~ 96 / dt^3, so dt^3 / 100 is a stable step for it
        rob, task = dataset.robot, dataset.task
        H = dataset.n_support_points
        costs = [CostCollision(rob, H, field=f, sigma_coll=1.0) for f in task.get_collision_fields()]
        weights = [step_coll] * len(costs)
        costs.append(CostGPTrajectory(rob, H, dt, sigma_gp=sigma_gp))
        weights.append(float(step_gp) if step_gp is not None else dt ** 3 / 100.0)
        self.step_coll, self.step_gp = weights[0], weights[-1]
        self.gp, self._prims = build_device_params(rob, dataset.env.dim, task.obstacle_cutoff_margin, None, None, costs, weights, True,
                                                   n_interp, clip_grad, max_grad_norm, device, identity_normalizer=True)
        self.D = 2 * rob.q_dim

    @torch.no_grad()
    def optimize(self, trajs: torch.Tensor, opt_iters: int = 500, return_iterations: bool = False):
        x = trajs.to(torch.float32).contiguous().clone()
        if not x.is_cuda:
            raise RuntimeError("GPMPOptimizer runs on the GPU (libmpdx guide kernel); there is no CPU fallback")
        B, H, D = x.shape
        hs, hg = x[:, 0].contiguous().clone(), x[:, -1].contiguous().clone()
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)   # the range test is bypassed by identity_normalizer
        lib, st = _lib.load(), _lib.current_stream()
        iters = [x.clone()] if return_iterations else None
        for _ in range(int(opt_iters)):
            _lib.check(lib.mpdx_guide_step(C.byref(self.gp), x.data_ptr(), None, hs.data_ptr(), hg.data_ptr(), flag.data_ptr(), None, B, B, H, D, st),
                       "mpdx_guide_step")
            if return_iterations:
                iters.append(x.clone())
        return (x, torch.stack(iters)) if return_iterations else x


class GPMP2:
    """GPMP2 (Mukadam et al., IJRR 2018) as Levenberg-Marquardt on the factor graph  GP prior + hinge collision factors, start and goal
    states fixed - what `GPMP2(**planner_params).optimize()` does at generate_trajectories.py:107-120 (un-vendored there).  One kernel
    launch per iteration for the whole batch (csrc/planner.hpp gpmp_lm_kernel); `oracle/gpmp.py` restates a step.

    The factor weights are this port's choice (the reference takes them from the un-vendored `env.get_gpmp2_params`): what the optimum
    depends on is sigma_obs / sigma_gp; on three start / goal draws of the narrow-passage environment (100 RRT-initialised trajectories each,
    500 iterations, gpurun_out/r04r/np_exp.txt) 2e-3 : 1 left 100 / 100 / 100 % of the trajectories collision free, 1e-3 : 1 (the round-3
    default) 91 / 100 / 99 %, stiffer ratios fewer still - the hinge factors' linearisation is only good near the margin."""

    def __init__(self, dataset: TrajectoryDataset, dt: float, sigma_gp: float = 1.0, sigma_obs: float = 2e-3, n_interp: int = 128,
                 lambda_init: float = 1e-2, lambda_up: float = 10.0, lambda_down: float = 0.2, lambda_min: float = 1e-7, lambda_max: float = 1e7,
                 step: float = 1.0, adaptive: bool = True, device="cuda"):
        rob, task = dataset.robot, dataset.task
        H = dataset.n_support_points
        costs = [CostCollision(rob, H, field=f, sigma_coll=1.0) for f in task.get_collision_fields()]
        costs.append(CostGPTrajectory(rob, H, dt, sigma_gp=sigma_gp))
        self.gp, self._prims = build_device_params(rob, dataset.env.dim, task.obstacle_cutoff_margin, None, None, costs, [1.0] * len(costs), True,
                                                   n_interp, False, 1.0, device, identity_normalizer=True)
        self.opts = _lib.GpmpOpts()
        self.opts.sigma_obs, self.opts.step, self.opts.adaptive = float(sigma_obs), float(step), int(bool(adaptive))
        self.opts.lambda_up, self.opts.lambda_down, self.opts.lambda_min, self.opts.lambda_max = float(lambda_up), float(lambda_down), float(lambda_min), float(lambda_max)
        self.lambda_init = float(lambda_init)
        self.D = 2 * rob.q_dim
        self.state = None

    @torch.no_grad()
    def optimize(self, trajs: torch.Tensor, opt_iters: int = 100, return_iterations: bool = False):
        x = trajs.to(torch.float32).contiguous().clone()
        if not x.is_cuda:
            raise RuntimeError("GPMP2 runs on the GPU (libmpdx); there is no CPU fallback")
        B, H, D = x.shape
        delta = torch.zeros_like(x)
        state = torch.zeros((B, 4), dtype=torch.float32, device=x.device)
        state[:, 0], state[:, 1] = 3.0e38, self.lambda_init
        lib, st = _lib.load(), _lib.current_stream()
        iters = [x.clone()] if return_iterations else None
        for it in range(int(opt_iters) + 1):   # the last call only judges the last proposal
            _lib.check(lib.mpdx_gpmp_step(C.byref(self.gp), C.byref(self.opts), x.data_ptr(), delta.data_ptr(), state.data_ptr(), B, H, D,
                                          1 if it < opt_iters else 0, st), "mpdx_gpmp_step")
            if return_iterations and it > 0:
                iters.append(x.clone())
        self.state = state    # [B, 4]: F, lambda, accepted steps, F of the last candidate
        return (x, torch.stack(iters)) if return_iterations else x


# ------------------------------------------------------------------------------------------------ the entry
def generate_collision_free_trajectories(env_id, robot_id, num_trajectories_per_context, results_dir, threshold_start_goal_pos=1.0,
                                         obstacle_cutoff_margin=0.03, n_tries=1000, rrt_max_time=300, gpmp_opt_iters=500,
                                         n_support_points=64, duration=5.0, tensor_args=None, debug=False, seed: int = 0,
                                         start_state_pos=None, goal_state_pos=None, rrt_step_size: Optional[float] = None):
    tensor_args = tensor_args or {"device": torch.device("cuda"), "dtype": torch.float32}
    dev = torch.device(tensor_args["device"])
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("generate_collision_free_trajectories needs an AMD GPU (no CPU fallback)")
    ds = TrajectoryDataset(env_id=env_id, robot_id=robot_id, n_support_points=n_support_points, obstacle_cutoff_margin=obstacle_cutoff_margin,
                           tensor_args=tensor_args)
    task, robot = ds.task, ds.robot
    gen = torch.Generator(device=dev).manual_seed(int(seed))
    # -------------------------------- start / goal (:52-64)
    if start_state_pos is None or goal_state_pos is None:
        start_state_pos = goal_state_pos = None
        for _ in range(n_tries):
            q_free = task.random_coll_free_q(n_samples=2, device=dev, generator=gen)
            if torch.linalg.norm(q_free[0] - q_free[1]) > threshold_start_goal_pos:
                start_state_pos, goal_state_pos = q_free[0], q_free[1]
                break
        if start_state_pos is None:
            raise ValueError("No collision free configuration was found")
    n = int(num_trajectories_per_context)
    dt = duration / n_support_points
    times = {}
    # -------------------------------- sample-based initialisation (:68-90)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = rrt_step_size or (0.1 if robot.q_dim <= 3 else 0.25)
    rrt = RRTConnectBatch(task, start_state_pos.to(dev), goal_state_pos.to(dev), n, step_size=step, generator=gen)
    deadline_iters = 6000
    used = rrt.grow(max_iters=deadline_iters)
    # path extraction + shortcutting + arc-length resampling on the device (an unsolved problem falls back to the straight line: the
    # optimiser may still repair it; it is reported as colliding otherwise)
    trajs0 = rrt.trajectories(n_support_points, dt)
    torch.cuda.synchronize()
    times["rrt_connect_s"] = time.perf_counter() - t0
    # -------------------------------- optimisation-based refinement (:92-120)
    t1 = time.perf_counter()
    opt = GPMP2(ds, dt, device=dev)
    trajs_last_iter, trajs_iters = opt.optimize(trajs0, opt_iters=gpmp_opt_iters, return_iterations=True)
    torch.cuda.synchronize()
    times["gpmp_s"] = time.perf_counter() - t1
    # -------------------------------- statistics and artefacts (:122-152)
    frac_free = task.compute_fraction_free_trajs(trajs_last_iter)
    intensity = task.compute_collision_intensity_trajs(trajs_last_iter)
    if debug:
        print("----------------STATISTICS----------------")
        print(f"rrt-connect: {int(rrt.done.sum())}/{n} solved in {used} iterations, {times['rrt_connect_s']:.3f} s; optimiser {times['gpmp_s']:.3f} s")
        print(f"percentage free trajs: {frac_free*100:.2f}\npercentage collision intensity {intensity*100:.2f}\nsuccess {task.compute_success_free_trajs(trajs_last_iter)}")
    coll, free = task.get_trajs_collision_and_free(trajs_last_iter)
    coll = torch.empty(0) if coll is None else coll
    free = torch.empty(0) if free is None else free
    if results_dir:
        os.makedirs(results_dir, exist_ok=True)
        torch.save(coll.cpu(), os.path.join(results_dir, "trajs-collision.pt"))
        torch.save(free.cpu(), os.path.join(results_dir, "trajs-free.pt"))
        results_data_dict = {"duration": duration, "n_support_points": n_support_points, "dt": dt,
                             "trajs_iters_coll": coll.unsqueeze(0).cpu() if coll.numel() else None,
                             "trajs_iters_free": free.unsqueeze(0).cpu() if free.numel() else None,
                             "times": times, "rrt_solved": int(rrt.done.sum()), "rrt_iterations": used}
        with open(os.path.join(results_dir, "results_data_dict.pickle"), "wb") as handle:
            pickle.dump(results_data_dict, handle, protocol=pickle.HIGHEST_PROTOCOL)
    generate_collision_free_trajectories.last = {"trajs_init": trajs0, "trajs_iters": trajs_iters, "times": times, "fraction_free": frac_free,
                                                 "collision_intensity": intensity, "rrt_solved": int(rrt.done.sum())}
    return len(coll), len(free)


def experiment(env_id: str = "EnvSpheres3D", robot_id: str = "RobotPanda", n_support_points: int = 64, duration: float = 5.0,
               threshold_start_goal_pos: float = 1.83, obstacle_cutoff_margin: float = 0.05, num_trajectories: int = 5, device: str = "cuda",
               debug: bool = True, seed: int = 0, results_dir: str = "data", **kwargs):
    """scripts/generate_data/generate_trajectories.py:170-246 with the same arguments: one start / goal context, `num_trajectories` trajectories, `metadata.yaml`
    next to `trajs-free.pt` / `trajs-collision.pt` / `results_data_dict.pickle` in `results_dir` (what the dataset loader walks: trajectories.py:84-123)."""
    import yaml
    if debug:
        torch.manual_seed(seed)   # fix_random_seed(seed)
    os.makedirs(results_dir, exist_ok=True)
    tensor_args = {"device": torch.device(device), "dtype": torch.float32}
    metadata = {"env_id": env_id, "robot_id": robot_id, "num_trajectories": num_trajectories}
    with open(os.path.join(results_dir, "metadata.yaml"), "w") as f:
        yaml.safe_dump(metadata, f)
    n_coll, n_free = generate_collision_free_trajectories(env_id, robot_id, num_trajectories, results_dir, threshold_start_goal_pos=threshold_start_goal_pos,
                                                          obstacle_cutoff_margin=obstacle_cutoff_margin, n_support_points=n_support_points, duration=duration,
                                                          tensor_args=tensor_args, debug=debug, seed=seed,
                                                          **{k: v for k, v in kwargs.items() if k in ("n_tries", "rrt_max_time", "gpmp_opt_iters", "rrt_step_size",
                                                                                                      "start_state_pos", "goal_state_pos")})
    metadata.update(num_trajectories_generated=n_coll + n_free, num_trajectories_generated_coll=n_coll, num_trajectories_generated_free=n_free)
    with open(os.path.join(results_dir, "metadata.yaml"), "w") as f:
        yaml.safe_dump(metadata, f)
    return n_coll, n_free


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="one start / goal context of the training set: python -m mpd_public_amd.generate_trajectories --env_id EnvSimple2D "
                                             "--robot_id RobotPointMass --num_trajectories 16 --results_dir data_trajectories/EnvSimple2D-RobotPointMass/0 --seed 0")
    ap.add_argument("--env_id", default="EnvSpheres3D")
    ap.add_argument("--robot_id", default="RobotPanda")
    ap.add_argument("--num_trajectories", type=int, default=5)
    ap.add_argument("--n_support_points", type=int, default=64)
    ap.add_argument("--threshold_start_goal_pos", type=float, default=None, help="default: 1.83 (Panda) / 1.0 (point mass), launch_generate_trajectories.py:13-16")
    ap.add_argument("--obstacle_cutoff_margin", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--results_dir", default="data")
    a = vars(ap.parse_args())
    if a["threshold_start_goal_pos"] is None:
        a["threshold_start_goal_pos"] = 1.83 if a["robot_id"] == "RobotPanda" else 1.0
    experiment(**a)
