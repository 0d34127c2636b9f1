"""The field normalisers and the slice of TrajectoryDataset the planning loop touches.

Reference: mpd/datasets/normalization.py:85-195 (LimitsNormalizer - the default, trajectories.py:26 - and the four others) and mpd/datasets/trajectories.py:196-237
(unnormalize_trajectories, get_hard_conditions).  Without a dataset directory the limits are given explicitly (synthetic,
SURVEY 8d).  With `base_dir=` the `trajs-free.pt` shards under it are loaded as the reference does (trajectories.py:84-110:
os.walk, one task id per shard, the `task` field = start/goal positions) and the limits come from the data
(normalization.py:144-167) - this is the training set of mpd_public_amd.trainer / train.py; the shards are what
`generate_trajectories.py` (the reference's, or mpd_public_amd.generate_trajectories) writes.
The in-loop unnormalise runs inside the HIP guide kernel; the methods below serve the one-off calls outside the loop
(hard conditions, un-normalising the returned chain) with plain torch ops.
"""
from __future__ import annotations

import torch

from . import synthetic as syn
from .planning import PlanningTask, make_env, make_robot


class LimitsNormalizer:
    """maps [xmin, xmax] to [-1, 1] (normalization.py:144-167); built from the limits (what the planning path holds) or, the reference's way,
    from a flattened field with `make_normalizer`"""
    kind = "limits"   # what the HIP guide kernel un-normalises with: limits + the whole-tensor range test

    def __init__(self, mins, maxs):
        self.mins = torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = torch.as_tensor(maxs, dtype=torch.float32)

    def to(self, device):
        self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def __call__(self, x):
        return self.normalize(x)

    def normalize(self, x):
        x = (x - self.mins) / (self.maxs - self.mins)
        return 2 * x - 1

    def unnormalize(self, x, eps=1e-4):
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clip(x, -1, 1)
        x = (x + 1) / 2.0
        return x * (self.maxs - self.mins) + self.mins


class SafeLimitsNormalizer(LimitsNormalizer):
    """LimitsNormalizer for data with a constant dimension (normalization.py:170-184): as soon as ONE dimension is constant the reference widens
    EVERY dimension's limits by eps (its loop subtracts from / adds to the whole vectors), once."""

    def __init__(self, mins, maxs, eps=1):
        super().__init__(mins, maxs)
        if eps != 0 and bool((self.mins == self.maxs).any()):
            self.mins, self.maxs = self.mins - eps, self.maxs + eps


class FixedLimitsNormalizer(LimitsNormalizer):
    """LimitsNormalizer with given limits instead of the data's (normalization.py:187-195)"""

    def __init__(self, mins, maxs, min=-1, max=1):
        super().__init__(mins, maxs)
        self.mins, self.maxs = torch.ones_like(self.mins) * min, torch.ones_like(self.maxs) * max


class Identity:
    """normalization.py:111-116.  The guide kernel skips its un-normalisation (`identity_normalizer`, as the baseline planners run it)."""
    kind = "identity"

    def __init__(self, mins=None, maxs=None):
        self.mins = None if mins is None else torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = None if maxs is None else torch.as_tensor(maxs, dtype=torch.float32)

    def to(self, device):
        if self.mins is not None:
            self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def __call__(self, x):
        return x

    def normalize(self, x):
        return x

    def unnormalize(self, x):
        return x


class GaussianNormalizer:
    """zero mean / unit variance per dimension (normalization.py:119-141; unbiased std).  The HIP guide kernel un-normalises with it as the reference does
    (x * stds + means, no range test: mpdx_guide_params.identity_normalizer == 2)."""
    kind = "gaussian"

    def __init__(self, means, stds, mins=None, maxs=None):
        self.means, self.stds = torch.as_tensor(means, dtype=torch.float32), torch.as_tensor(stds, dtype=torch.float32)
        self.mins = None if mins is None else torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = None if maxs is None else torch.as_tensor(maxs, dtype=torch.float32)
        self.z = 1

    def to(self, device):
        self.means, self.stds = self.means.to(device), self.stds.to(device)
        if self.mins is not None:
            self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def __call__(self, x):
        return self.normalize(x)

    def normalize(self, x):
        return (x - self.means) / self.stds

    def unnormalize(self, x):
        return x * self.stds + self.means


NORMALIZERS = {c.__name__: c for c in (Identity, GaussianNormalizer, LimitsNormalizer, SafeLimitsNormalizer, FixedLimitsNormalizer)}


def make_normalizer(normalizer, X, **kw):
    """What DatasetNormalizer.__init__ does per field (normalization.py:14-22: `eval(normalizer)(X)` on the flattened field [N, dim]):
    `normalizer` is one of the five class names (or classes) of the reference's module."""
    name = normalizer if isinstance(normalizer, str) else normalizer.__name__
    if name not in NORMALIZERS:
        raise NameError(f"name {name!r} is not defined")   # (the reference's eval() of an unknown name)
    X = torch.as_tensor(X, dtype=torch.float32)
    X = X.reshape(-1, X.shape[-1])
    mins, maxs = X.min(dim=0).values, X.max(dim=0).values   # Normalizer.__init__ (normalization.py:90-93)
    if name == "GaussianNormalizer":
        return GaussianNormalizer(X.mean(dim=0), X.std(dim=0), mins, maxs)
    return NORMALIZERS[name](mins, maxs, **kw)


class TrajectoryDataset:
    """`dataset` as inference.py uses it: .env .robot .task .n_support_points .state_dim .threshold_start_goal_pos,
    normalize/unnormalize_trajectories, get_hard_conditions."""

    field_key_traj = "traj"

    def __init__(self, env_id="EnvDense2D", robot_id="RobotPointMass", n_support_points=64, include_velocity=True,
                 obstacle_cutoff_margin=0.05, use_extra_objects=True, tensor_args=None, base_dir=None, normalizer="LimitsNormalizer", **kw):
        self.tensor_args = tensor_args or {"device": "cpu", "dtype": torch.float32}
        self.env, self.robot = make_env(env_id), make_robot(robot_id)
        self.task = PlanningTask(self.env, self.robot, obstacle_cutoff_margin=obstacle_cutoff_margin,
                                 use_extra_objects=use_extra_objects, tensor_args=self.tensor_args)
        self.n_support_points, self.include_velocity = n_support_points, include_velocity
        self.state_dim = self.robot.q_dim * (2 if include_velocity else 1)
        # `normalizer`: one of the reference's five class names (trajectories.py:26,78).  Without a dataset directory the limits are this package's
        # synthetic ones (a GaussianNormalizer needs data: base_dir)
        self.normalizer_name = normalizer if isinstance(normalizer, str) else normalizer.__name__
        if self.normalizer_name not in NORMALIZERS:
            raise NameError(f"name {self.normalizer_name!r} is not defined")
        mins, maxs = syn.limits_for(robot_id)
        mins, maxs = mins[: self.state_dim], maxs[: self.state_dim]
        if self.normalizer_name == "GaussianNormalizer":
            if base_dir is None:
                raise ValueError("normalizer='GaussianNormalizer' takes its statistics from the data: give base_dir")
            self.normalizer = None
        else:
            self.normalizer = NORMALIZERS[self.normalizer_name](mins, maxs).to(self.tensor_args["device"])
        self.threshold_start_goal_pos = 1.0 if self.robot.q_dim <= 3 else 1.83  # launch_generate_trajectories.py:13-16
        self.fields = {}
        self.map_task_id_to_trajectories_id, self.map_trajectory_id_to_task_id = {}, {}
        if base_dir is not None:
            self.load_trajectories(base_dir)

    # ---- the training set (trajectories.py:84-123, 150-172)
    def load_trajectories(self, base_dir):
        import os
        import numpy as np
        dev = self.tensor_args["device"]
        shards, task_id, n = [], 0, 0
        for current_dir, subdirs, files in sorted(os.walk(base_dir, topdown=True)):
            if "trajs-free.pt" in files:
                tr = torch.load(os.path.join(current_dir, "trajs-free.pt"), map_location=dev).to(torch.float32)
                if tr.numel() == 0:
                    continue
                idx = n + np.arange(len(tr))
                self.map_task_id_to_trajectories_id[task_id] = idx
                for j in idx:
                    self.map_trajectory_id_to_task_id[int(j)] = task_id
                task_id += 1
                n += len(tr)
                shards.append(tr)
        if not shards:
            raise FileNotFoundError(f"no trajs-free.pt under {base_dir}")
        trajs_free = torch.cat(shards)
        pos = self.robot.get_position(trajs_free)
        trajs = trajs_free if self.include_velocity else pos
        if trajs.shape[-1] != self.state_dim:
            raise ValueError(f"trajectories have state dim {trajs.shape[-1]}, expected {self.state_dim}")
        self.fields["traj"] = trajs
        self.fields["task"] = torch.cat((pos[..., 0, :], pos[..., -1, :]), dim=-1)
        self.n_trajs, self.n_support_points = trajs.shape[0], trajs.shape[1]
        self.trajectory_dim = (self.n_support_points, self.state_dim)
        flat = trajs.reshape(-1, self.state_dim)
        self.normalizer = make_normalizer(self.normalizer_name, flat).to(dev)   # normalization.py:14-22,90-93
        tflat = self.fields["task"]
        self._task_normalizer = make_normalizer(self.normalizer_name, tflat).to(dev)
        self.fields["traj_normalized"] = self.normalizer.normalize(trajs)
        self.fields["task_normalized"] = self._task_normalizer.normalize(tflat)

    def __len__(self):
        return self.fields["traj"].shape[0] if "traj" in self.fields else 0

    def __getitem__(self, index):
        traj_normalized = self.fields["traj_normalized"][index]
        data = {"traj_normalized": traj_normalized, "task_normalized": self.fields["task_normalized"][index]}
        data["hard_conds"] = self.get_hard_conditions(traj_normalized, horizon=len(traj_normalized))
        return data

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
