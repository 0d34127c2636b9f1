"""LimitsNormalizer and the slice of TrajectoryDataset the planning loop touches.

Reference: mpd/datasets/normalization.py:144-167 (LimitsNormalizer) and mpd/datasets/trajectories.py:196-237
(unnormalize_trajectories, get_hard_conditions).  Loading the authors' `trajs-free.pt` shards (trajectories.py:45-80,
needs the Google-Drive dataset + gitpython) is out of scope: limits are given explicitly (synthetic).
The in-loop unnormalise runs inside the HIP guide kernel; the methods below serve the one-off calls outside the loop
(hard conditions, un-normalising the returned chain) with plain torch ops.
"""
from __future__ import annotations

import torch

from . import synthetic as syn
from .planning import PlanningTask, make_env, make_robot


class LimitsNormalizer:
    """maps [xmin, xmax] to [-1, 1]"""

    def __init__(self, mins, maxs):
        self.mins = torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = torch.as_tensor(maxs, dtype=torch.float32)

    def to(self, device):
        self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def normalize(self, x):
        x = (x - self.mins) / (self.maxs - self.mins)
        return 2 * x - 1

    def unnormalize(self, x, eps=1e-4):
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clip(x, -1, 1)
        x = (x + 1) / 2.0
        return x * (self.maxs - self.mins) + self.mins


class TrajectoryDataset:
    """`dataset` as inference.py uses it: .env .robot .task .n_support_points .state_dim .threshold_start_goal_pos,
    normalize/unnormalize_trajectories, get_hard_conditions."""

    field_key_traj = "traj"

    def __init__(self, env_id="EnvDense2D", robot_id="RobotPointMass", n_support_points=64, include_velocity=True,
                 obstacle_cutoff_margin=0.05, use_extra_objects=True, tensor_args=None, **kw):
        self.tensor_args = tensor_args or {"device": "cpu", "dtype": torch.float32}
        self.env, self.robot = make_env(env_id), make_robot(robot_id)
        self.task = PlanningTask(self.env, self.robot, obstacle_cutoff_margin=obstacle_cutoff_margin,
                                 use_extra_objects=use_extra_objects, tensor_args=self.tensor_args)
        self.n_support_points, self.include_velocity = n_support_points, include_velocity
        self.state_dim = self.robot.q_dim * (2 if include_velocity else 1)
        mins, maxs = syn.limits_for(robot_id)
        self.normalizer = LimitsNormalizer(mins[: self.state_dim], maxs[: self.state_dim]).to(self.tensor_args["device"])
        self.threshold_start_goal_pos = 1.0 if self.robot.q_dim <= 3 else 1.83  # launch_generate_trajectories.py:13-16

    def normalize_trajectories(self, x):
        return self.normalizer.normalize(x)

    def unnormalize_trajectories(self, x):
        return self.normalizer.unnormalize(x)

    def get_hard_conditions(self, traj, horizon=None, normalize=False):
        start_pos, goal_pos = self.robot.get_position(traj[0]), self.robot.get_position(traj[-1])
        if self.include_velocity:  # zero velocity at both ends (trajectories.py:219-223)
            start = torch.cat((start_pos, torch.zeros_like(start_pos)), dim=-1)
            goal = torch.cat((goal_pos, torch.zeros_like(goal_pos)), dim=-1)
        else:
            start, goal = start_pos, goal_pos
        if normalize:
            start, goal = self.normalizer.normalize(start), self.normalizer.normalize(goal)
        horizon = horizon or self.n_support_points
        return {0: start, horizon - 1: goal}
