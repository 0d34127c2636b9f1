"""Host-side descriptors of the planning problem: environments, robots, the planning task and the cost terms.

In the reference these come from the un-vendored `torch_robotics` / `mp_baselines` submodules (empty in the reference
tree): envs `EnvSimple2D / EnvDense2D / EnvNarrowPassageDense2D / EnvSpheres3D`, robots `RobotPointMass / RobotPanda`,
`PlanningTask`, and `CostCollision / CostGPTrajectory / CostComposite` (scripts/inference/inference.py:14,107-123,
188-225).  Here they are thin DESCRIPTORS with the reference's constructor signatures; the arithmetic runs in the HIP
guide kernel (csrc/guide.hpp).  Geometry is synthetic and formula-defined - the authors'
environments are not in the reference tree.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import synthetic as syn
from . import _lib

# ------------------------------------------------------------------------------------------------ geometry


@dataclass
class ObjectSet:
    """sphere + axis-aligned box primitives; rows padded to 3-D (z = 0 in 2-D)."""
    sphere_centers: np.ndarray  # [ns, 3]
    sphere_radii: np.ndarray    # [ns]
    box_centers: np.ndarray     # [nb, 3]
    box_half: np.ndarray        # [nb, 3]

    @staticmethod
    def empty():
        z3, z1 = np.zeros((0, 3), np.float32), np.zeros((0,), np.float32)
        return ObjectSet(z3, z1, z3.copy(), z3.copy())

    def prim_floats(self):
        sp = np.concatenate([self.sphere_centers, self.sphere_radii[:, None]], 1).astype(np.float32).reshape(-1)
        bx = np.concatenate([self.box_centers, self.box_half], 1).astype(np.float32).reshape(-1)
        return sp, bx


def _pad3(a, dim):
    a = np.asarray(a, np.float32).reshape(-1, dim)
    return np.concatenate([a, np.zeros((a.shape[0], 3 - dim), np.float32)], 1)


class Env:
    def __init__(self, name: str, dim: int, fixed: ObjectSet, extra: ObjectSet, limits=(-1.0, 1.0)):
        self.name, self.dim, self.obj_fixed, self.obj_extra = name, dim, fixed, extra
        self.limits = (np.full(dim, limits[0], np.float32), np.full(dim, limits[1], np.float32))


def _spheres(tag, n, dim, rlo, rhi, lo=-0.9, hi=0.9):
    c = syn.hash_uniform(f"{tag}/centers", n * dim, lo, hi).reshape(n, dim)
    r = syn.hash_uniform(f"{tag}/radii", n, rlo, rhi) if rhi > rlo else np.full(n, rlo)
    return _pad3(c, dim), r.astype(np.float32)


def _boxes(tag, n, dim, half, lo=-0.9, hi=0.9):
    c = syn.hash_uniform(f"{tag}/centers", n * dim, lo, hi).reshape(n, dim)
    h = np.full((n, dim), half, np.float32)
    hp = np.concatenate([h, np.full((n, 3 - dim), 1.0, np.float32)], 1)  # 2-D boxes are unbounded along the unused axis
    return _pad3(c, dim), hp


def make_env(env_id: str) -> Env:
    """Synthetic stand-ins for the reference's environments."""
    if env_id == "EnvSimple2D":
        sc, sr = _spheres("simple2d/s", 8, 2, 0.1, 0.2)
        bc, bh = _boxes("simple2d/b", 2, 2, 0.1)
        ec, er = _spheres("simple2d/es", 2, 2, 0.1, 0.15)
        ebc, ebh = _boxes("simple2d/eb", 2, 2, 0.08)
        return Env(env_id, 2, ObjectSet(sc, sr, bc, bh), ObjectSet(ec, er, ebc, ebh))
    if env_id in ("EnvDense2D", "EnvNarrowPassageDense2D"):
        sc, sr = _spheres("dense2d/s", 20, 2, 0.125, 0.125)
        bc, bh = _boxes("dense2d/b", 6, 2, 0.1)
        if env_id == "EnvNarrowPassageDense2D":  # two wall boxes leaving a 0.1 gap at x = 0
            wc = _pad3([[0.0, 0.525], [0.0, -0.525]], 2)
            wh = np.array([[0.05, 0.475, 1.0], [0.05, 0.475, 1.0]], np.float32)
            bc, bh = np.concatenate([bc, wc]), np.concatenate([bh, wh])
        ec, er = _spheres("dense2d/es", 2, 2, 0.1, 0.125)
        ebc, ebh = _boxes("dense2d/eb", 2, 2, 0.08)
        return Env(env_id, 2, ObjectSet(sc, sr, bc, bh), ObjectSet(ec, er, ebc, ebh))
    if env_id == "EnvSpheres3D":
        sc, sr = _spheres("spheres3d/s", 15, 3, 0.15, 0.15, -0.8, 0.8)
        sc[:, 2] = 0.2 + 0.8 * (sc[:, 2] + 0.8) / 1.6  # keep them in the arm's workspace (z in [0.2, 1.0])
        ec, er = _spheres("spheres3d/es", 2, 3, 0.12, 0.15, -0.6, 0.6)
        ec[:, 2] = 0.3 + 0.5 * (ec[:, 2] + 0.6) / 1.2
        z = ObjectSet.empty()
        env = Env(env_id, 3, ObjectSet(sc, sr, z.box_centers, z.box_half), ObjectSet(ec, er, z.box_centers.copy(), z.box_half.copy()))
        env.limits = (np.array([-1.0, -1.0, -0.1], np.float32), np.array([1.0, 1.0, 1.5], np.float32))
        return env
    raise NotImplementedError(env_id)


# ------------------------------------------------------------------------------------------------ robots


class RobotPointMass:
    name = "RobotPointMass"

    def __init__(self, q_dim=2, link_margin=0.01, tensor_args=None, **kw):
        self.q_dim, self.link_margin, self.dt = q_dim, link_margin, None
        self.robot_id = _lib.ROBOT_POINTMASS

    def get_position(self, x):
        return x[..., : self.q_dim]

    def get_velocity(self, x):
        return x[..., self.q_dim: 2 * self.q_dim]


class RobotPanda:
    name = "RobotPanda"

    def __init__(self, tensor_args=None, **kw):
        self.q_dim, self.link_margin, self.dt = 7, 0.0, None
        self.robot_id = _lib.ROBOT_PANDA

    get_position = RobotPointMass.get_position
    get_velocity = RobotPointMass.get_velocity


def make_robot(robot_id: str):
    if robot_id == "RobotPointMass":
        return RobotPointMass(2)
    if robot_id == "RobotPointMass3D":
        return RobotPointMass(3)
    if robot_id == "RobotPanda":
        return RobotPanda()
    raise NotImplementedError(robot_id)


# ------------------------------------------------------------------------------------------------ collision fields / task


@dataclass
class CollisionField:
    kind: int                       # _lib.FIELD_*
    objects: Optional[ObjectSet] = None
    ws_min: Optional[np.ndarray] = None
    ws_max: Optional[np.ndarray] = None
    name: str = ""


class PlanningTask:
    """The attribute / method surface inference.py reads from `task` (inference.py:161,191-193,288-297)."""

    def __init__(self, env: Env, robot, obstacle_cutoff_margin=0.05, use_extra_objects=True, tensor_args=None, **kw):
        self.env, self.robot, self.obstacle_cutoff_margin = env, robot, obstacle_cutoff_margin
        self.tensor_args = tensor_args or {"device": "cpu", "dtype": torch.float32}
        self.ws_min, self.ws_max = env.limits
        self.df_collision_objects = CollisionField(_lib.FIELD_OBJECTS, objects=env.obj_fixed, name="objects")
        self.df_collision_extra_objects = CollisionField(_lib.FIELD_OBJECTS, objects=env.obj_extra, name="extra_objects") if use_extra_objects else None
        self.df_collision_ws_boundaries = CollisionField(_lib.FIELD_WORKSPACE, ws_min=self.ws_min, ws_max=self.ws_max, name="workspace")
        self.df_collision_self = CollisionField(_lib.FIELD_SELF, name="self") if robot.name == "RobotPanda" else None

    def get_collision_fields(self) -> List[CollisionField]:
        out = [self.df_collision_self, self.df_collision_objects, self.df_collision_ws_boundaries, self.df_collision_extra_objects]
        return [f for f in out if f is not None]

    def get_collision_fields_extra_objects(self) -> List[CollisionField]:
        return [self.df_collision_extra_objects]


# ------------------------------------------------------------------------------------------------ cost descriptors


class CostCollision:
    def __init__(self, robot, n_support_points, field=None, sigma_coll=1.0, tensor_args=None, **kw):
        if sigma_coll != 1.0:
            raise NotImplementedError("sigma_coll != 1 (inference.py:201 always passes 1.0)")
        self.robot, self.n_support_points, self.field = robot, n_support_points, field


class CostGPTrajectory:
    def __init__(self, robot, n_support_points, dt, sigma_gp=1.0, tensor_args=None, **kw):
        self.robot, self.n_support_points, self.dt, self.sigma_gp = robot, n_support_points, float(dt), float(sigma_gp)


class CostComposite:
    def __init__(self, robot, n_support_points, cost_list, weights_cost_l=None, tensor_args=None, **kw):
        self.robot, self.n_support_points = robot, n_support_points
        self.cost_l = list(cost_list)
        self.weight_cost_l = list(weights_cost_l) if weights_cost_l is not None else [1.0] * len(self.cost_l)
        if len(self.cost_l) != len(self.weight_cost_l):
            raise ValueError("one weight per cost term")
