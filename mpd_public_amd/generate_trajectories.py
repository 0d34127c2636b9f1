"""generate_collision_free_trajectories() - drop-in for scripts/generate_data/generate_trajectories.py:20-169 (dataset generation
with the reference's baseline planners: RRT-Connect initialisation + GPMP2 optimisation = `HybridPlanner`, :66-120).

Same arguments, same artefacts (`trajs-free.pt`, `trajs-collision.pt`, `results_data_dict.pickle` with the same keys, :131-152),
same return value (number of colliding / collision-free trajectories).  It doubles as the like-for-like PLANNER BASELINE of the
diffusion sampler on the same box.

The reference's planners live in the un-vendored `mp_baselines` submodule (empty in /root/reference: PARITY UNPINNED, as for the
guide's costs - DESIGN.md section 5).  What is built here, MI355X-first:

  * RRT-Connect (Kuffner & LaValle 2000), BATCHED: the n trajectories of a context are n independent bidirectional trees grown in
    lock-step as device tensors [n, 2, max_nodes, q]; nearest-neighbour search, steering and bookkeeping are torch ops on the GPU,
    and every edge is collision-checked by the HIP metrics kernel (`mpdx_traj_metrics`: an edge is a 2-waypoint trajectory checked
    on `n_edge_checks` interpolated points against the task's collision fields) - one launch per extension for the whole batch.
    The reference's `MultiSampleBasedPlanner` runs its n RRTs one after the other in Python (:85-90).
  * Trajectory optimisation on the GPMP2 objective (Mukadam et al. 2018: constant-velocity GP prior + hinge collision factors on the
    interpolated trajectory): gradient descent with the HIP guide kernel (`csrc/guide.hpp`, hand-derived gradients, per-waypoint norm
    clip as trust region, start/goal hard-conditioned) in raw robot units (`identity_normalizer`), `opt_iters` launches of ~6-16 us.
    This is NOT GPMP2's Gauss-Newton step (that needs the un-vendored factor-graph code to pin); it minimises the same cost.

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
    """n independent RRT-Connect problems (one start/goal pair, n samples - or per-problem starts/goals) grown in lock-step."""

    def __init__(self, task, start: torch.Tensor, goal: torch.Tensor, n: int, step_size: float = 0.1, max_nodes: int = 2048,
                 n_edge_checks: int = 16, generator: Optional[torch.Generator] = None):
        dev = start.device
        if dev.type != "cuda":
            raise RuntimeError("RRTConnectBatch runs on the GPU (libmpdx collision kernel); there is no CPU fallback")
        self.task, self.n, self.step, self.M, self.nchk, self.gen = task, n, float(step_size), int(max_nodes), n_edge_checks, generator
        q = start.shape[-1]
        self.q = q
        self.lo, self.hi = task.q_limits(dev)
        s = start.reshape(1, q).expand(n, q) if start.dim() == 1 else start
        g = goal.reshape(1, q).expand(n, q) if goal.dim() == 1 else goal
        self.nodes = torch.zeros((n, 2, self.M, q), device=dev)          # tree 0 grows from the start, tree 1 from the goal
        self.parent = torch.full((n, 2, self.M), -1, dtype=torch.long, device=dev)
        self.count = torch.ones((n, 2), dtype=torch.long, device=dev)
        self.nodes[:, 0, 0], self.nodes[:, 1, 0] = s, g
        self.done = torch.zeros(n, dtype=torch.bool, device=dev)
        self.link = torch.full((n, 2), -1, dtype=torch.long, device=dev)  # node indices (tree 0, tree 1) where the trees met
        self.ar = torch.arange(n, device=dev)

    def _nearest(self, tree: int, q: torch.Tensor) -> torch.Tensor:
        d = torch.linalg.norm(self.nodes[:, tree] - q[:, None, :], dim=-1)                      # [n, M]
        d = d.masked_fill(torch.arange(self.M, device=q.device)[None, :] >= self.count[:, tree, None], float("inf"))
        return d.argmin(dim=1)

    def _steer(self, qn: torch.Tensor, qt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        d = qt - qn
        dist = torch.linalg.norm(d, dim=-1, keepdim=True)
        reach = dist[:, 0] <= self.step
        return torch.where(reach[:, None], qt, qn + d * (self.step / dist.clamp_min(1e-12))), reach

    def _add(self, tree: int, qnew: torch.Tensor, par: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        mask = mask & (self.count[:, tree] < self.M)
        idx = self.count[:, tree].clamp_max(self.M - 1)
        rows = self.ar[mask]
        self.nodes[rows, tree, idx[mask]] = qnew[mask]
        self.parent[rows, tree, idx[mask]] = par[mask]
        self.count[:, tree] += mask.long()
        return idx

    def grow(self, max_iters: int = 4000, max_connect_steps: int = 64) -> int:
        """Returns the number of iterations used.  One iteration = extend the active tree towards a random sample, then
        connect the other tree greedily towards the new node (alternating trees)."""
        dev = self.nodes.device
        it = 0
        for it in range(1, max_iters + 1):
            ta = it & 1
            tb = 1 - ta
            live = ~self.done
            qr = self.lo + (self.hi - self.lo) * torch.rand((self.n, self.q), device=dev, generator=self.gen)
            ia = self._nearest(ta, qr)
            qn = self.nodes[self.ar, ta, ia]
            qnew, _ = self._steer(qn, qr)
            ok = edges_free(self.task, qn, qnew, self.nchk) & live
            inew = self._add(ta, qnew, ia, ok)
            # connect: walk the other tree from its nearest node towards qnew until blocked or there
            ib = self._nearest(tb, qnew)
            cur = self.nodes[self.ar, tb, ib]
            cur_idx = ib
            active = ok.clone()
            for _ in range(max_connect_steps):
                if not bool(active.any()):
                    break
                nxt, reach = self._steer(cur, qnew)
                free = edges_free(self.task, cur, nxt, self.nchk) & active
                arrived = free & reach
                # a reached target is the SAME configuration as qnew: do not duplicate it, just record the link
                add_mask = free & ~reach
                k = self._add(tb, nxt, cur_idx, add_mask)
                self.link[arrived, ta] = inew[arrived]
                self.link[arrived, tb] = cur_idx[arrived]
                self.done |= arrived
                cur = torch.where(add_mask[:, None], nxt, cur)
                cur_idx = torch.where(add_mask, k, cur_idx)
                active = add_mask
            if bool(self.done.all()):
                break
        return it

    def paths(self) -> List[Optional[torch.Tensor]]:
        """Per problem: [n_nodes, q] configurations from start to goal (None if the trees did not meet)."""
        nodes, parent, link, done = self.nodes.cpu(), self.parent.cpu(), self.link.cpu(), self.done.cpu()
        out: List[Optional[torch.Tensor]] = []
        for i in range(self.n):
            if not bool(done[i]):
                out.append(None)
                continue
            branch = []
            for tree in (0, 1):
                seq, k = [], int(link[i, tree])
                while k >= 0:
                    seq.append(nodes[i, tree, k])
                    k = int(parent[i, tree, k])
                branch.append(seq)
            path = list(reversed(branch[0])) + branch[1]      # start ... meeting node | other tree's branch ... goal
            out.append(torch.stack(path))
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
    paths = rrt.paths()
    line = torch.stack([start_state_pos.cpu(), goal_state_pos.cpu()])
    init = []
    for p in paths:   # an unsolved problem falls back to the straight line (the optimiser may still repair it; it is reported as colliding otherwise)
        p = shortcut_path(task, p) if p is not None else line
        init.append(resample_path(p, n_support_points, dt))
    trajs0 = torch.stack(init).to(dev)
    torch.cuda.synchronize()
    times["rrt_connect_s"] = time.perf_counter() - t0
    # -------------------------------- optimisation-based refinement (:92-120)
    t1 = time.perf_counter()
    opt = GPMPOptimizer(ds, dt, device=dev)
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
