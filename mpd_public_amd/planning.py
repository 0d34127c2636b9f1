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

    # ---- collision checking / metrics (inference.py:161,288-297); arithmetic in csrc/guide.hpp::traj_metrics_kernel
    def _params(self, device):
        if getattr(self, "_gp", None) is None or self._gp_prims.device != torch.device(device):
            from .guides import build_device_params
            costs = [CostCollision(self.robot, 64, field=f) for f in self.get_collision_fields()]
            self._gp, self._gp_prims = build_device_params(self.robot, self.env.dim, self.obstacle_cutoff_margin, None, None, costs,
                                                           [1.0] * len(costs), True, 128, True, 1.0, device)
        return self._gp

    def trajectory_metrics(self, trajs, n_check=None, return_mask=False):
        """trajs: UNNORMALISED [B,H,D] on the GPU -> float tensor [B,4]: (#colliding waypoints, path length, smoothness,
        #waypoints checked); with return_mask also the per-waypoint collision flags [B, n_check] (bool) the count is made of."""
        import ctypes as C
        trajs = trajs.to(torch.float32).contiguous()
        if not trajs.is_cuda:
            raise RuntimeError("trajectory metrics run on the GPU (libmpdx); there is no CPU fallback")
        B, H, D = trajs.shape
        n_check = int(n_check or 4 * H)
        out = torch.empty((B, 4), dtype=torch.float32, device=trajs.device)
        mask = torch.empty((B, n_check), dtype=torch.uint8, device=trajs.device) if return_mask else None
        gp = self._params(trajs.device)
        _lib.check(_lib.load().mpdx_traj_metrics_mask(C.byref(gp), trajs.data_ptr(), out.data_ptr(), mask.data_ptr() if return_mask else None,
                                                      n_check, B, H, D, _lib.current_stream()), "mpdx_traj_metrics_mask")
        return (out, mask.bool()) if return_mask else out

    def get_trajs_collision_and_free(self, trajs, return_indices=False, **kw):
        m = self.trajectory_metrics(trajs)
        coll = m[:, 0] > 0
        idx_c, idx_f = torch.nonzero(coll).flatten(), torch.nonzero(~coll).flatten()
        tc = trajs[idx_c] if idx_c.numel() else None
        tf = trajs[idx_f] if idx_f.numel() else None
        if return_indices:
            return tc, idx_c, tf, idx_f, None
        return tc, tf

    def compute_fraction_free_trajs(self, trajs, **kw):
        m = self.trajectory_metrics(trajs)
        return float((m[:, 0] == 0).float().mean())

    def compute_collision_intensity_trajs(self, trajs, **kw):
        m = self.trajectory_metrics(trajs)
        return float((m[:, 0] / m[:, 3]).mean())

    def compute_success_free_trajs(self, trajs, **kw):
        return int(self.compute_fraction_free_trajs(trajs) > 0)

    def random_coll_free_q(self, n_samples=1, max_tries=1000, device="cuda", generator=None):
        """uniform collision-free configurations (inference.py:161)."""
        lo, hi = self.q_limits(device)
        got = []
        for _ in range(max_tries):
            q = lo + (hi - lo) * torch.rand((4 * n_samples, self.robot.q_dim), device=device, generator=generator)
            traj = torch.cat([q, torch.zeros_like(q)], -1)[:, None, :].expand(-1, 2, -1).contiguous()
            free = self.trajectory_metrics(traj, n_check=2)[:, 0] == 0
            got.append(q[free])
            if sum(g.shape[0] for g in got) >= n_samples:
                return torch.cat(got)[:n_samples]
        raise ValueError("No collision free configuration was found")

    def q_limits(self, device="cpu"):
        lo, hi = syn.limits_for(self.robot.name if self.robot.q_dim != 3 else "RobotPointMass3D")
        qd = self.robot.q_dim
        if self.robot.name == "RobotPointMass":
            lo, hi = np.concatenate([self.ws_min, lo[qd:]]), np.concatenate([self.ws_max, hi[qd:]])
        return torch.tensor(lo[:qd], device=device), torch.tensor(hi[:qd], device=device)


def task_from_torch_robotics(tr_task, tensor_args=None, use_extra_objects=True) -> PlanningTask:
    """Adapter: a REAL `torch_robotics.tasks.tasks.PlanningTask` (what `inference.py:161,181,191-201` builds) -> this package's `PlanningTask`, i.e. the
    primitive tables the guide / metrics kernels read.  torch_robotics is an empty submodule in the reference checkout (`deps/torch_robotics`,
    `.gitmodules:7-9`), so the attribute names below are the published package's AS RECALLED and cannot be exercised here against the real classes
    (tests/test_adapter_cpu.py drives it with stand-in objects of the same shape); every attribute read is listed, and a missing one raises
    AttributeError naming it instead of guessing:

      tr_task.env                      EnvBase:  .name (str, optional), .dim (2 | 3), .limits ([2, dim] tensor: workspace min / max)
      tr_task.env.obj_fixed_list       [ObjectField]  - the environment's own obstacles
      tr_task.env.obj_extra_list       [ObjectField] or None - `use_extra_objects=True` obstacles (inference.py:109)
        ObjectField.fields             [primitive fields]; ObjectField.pos (optional [dim] offset added to every centre; rotations are NOT supported: a
                                       non-identity ObjectField.ori raises)
          MultiSphereField             .centers [n, dim], .radii [n]
          MultiBoxField                .centers [n, dim], .sizes [n, dim] (FULL edge lengths; halved here) - or .half_sizes [n, dim]
      tr_task.robot                    .name ('RobotPointMass' | 'RobotPanda' | ...), .q_dim; RobotPointMass: .link_margins_for_object_collision_checking
                                       ([margin]) or .link_margin
      tr_task.obstacle_cutoff_margin   float (inference.py:110)

    Signed-distance GRIDS (`GridMapSDF`) and meshes have no primitive form: they raise NotImplementedError (the kernels scan sphere / box tables).
    The arithmetic on the tables (hinge on the SDF, FK, interpolation) is this package's restatement (parity unpinned)."""
    def need(obj, *names):
        for n in names:
            if hasattr(obj, n) and getattr(obj, n) is not None:
                return getattr(obj, n)
        raise AttributeError(f"{type(obj).__name__} has none of the attributes {names} the adapter reads (see task_from_torch_robotics.__doc__)")

    def arr(v, cols=None):
        a = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, np.float32)
        return a.reshape(-1, cols) if cols else a.reshape(-1)

    env_t = need(tr_task, "env")
    dim = int(need(env_t, "dim"))
    if dim not in (2, 3):
        raise NotImplementedError(f"workspace dimension {dim}")

    def object_set(obj_list):
        sc, sr, bc, bh = [], [], [], []
        for obj in (obj_list or []):
            ori = getattr(obj, "ori", None)
            if ori is not None:
                o = arr(ori)
                ident = (o.size == 4 and abs(abs(o[0]) - 1.0) < 1e-6 and np.abs(o[1:]).max() < 1e-6) or (o.size == 9 and np.abs(o.reshape(3, 3) - np.eye(3)).max() < 1e-6) \
                    or np.abs(o).max() < 1e-12
                if not ident:
                    raise NotImplementedError("rotated ObjectField: the primitive tables hold axis-aligned boxes")
            off = arr(obj.pos)[:dim] if getattr(obj, "pos", None) is not None else np.zeros(dim, np.float32)
            for f in need(obj, "fields"):
                kind = type(f).__name__
                if hasattr(f, "radii"):
                    c = arr(need(f, "centers"), dim) + off
                    sc.append(_pad3(c, dim)); sr.append(arr(f.radii))
                elif hasattr(f, "sizes") or hasattr(f, "half_sizes"):
                    c = arr(need(f, "centers"), dim) + off
                    h = arr(f.half_sizes, dim) if getattr(f, "half_sizes", None) is not None else 0.5 * arr(f.sizes, dim)
                    hp = np.concatenate([h, np.full((h.shape[0], 3 - dim), 1.0, np.float32)], 1)   # 2-D boxes are unbounded along the unused axis
                    bc.append(_pad3(c, dim)); bh.append(hp)
                else:
                    raise NotImplementedError(f"{kind}: only sphere (.centers, .radii) and box (.centers, .sizes) primitive fields have a table form")
        z = ObjectSet.empty()
        return ObjectSet(np.concatenate(sc) if sc else z.sphere_centers, np.concatenate(sr) if sr else z.sphere_radii,
                         np.concatenate(bc) if bc else z.box_centers, np.concatenate(bh) if bh else z.box_half)

    env = Env(str(getattr(env_t, "name", type(env_t).__name__)), dim, object_set(need(env_t, "obj_fixed_list")),
              object_set(getattr(env_t, "obj_extra_list", None)))
    lim = arr(need(env_t, "limits"), dim)
    env.limits = (lim[0].copy(), lim[1].copy())
    rob_t = need(tr_task, "robot")
    rname = str(getattr(rob_t, "name", type(rob_t).__name__))
    if "Panda" in rname:
        robot = RobotPanda()
    elif "PointMass" in rname:
        margin = getattr(rob_t, "link_margins_for_object_collision_checking", None)
        margin = float(arr(margin)[0]) if margin is not None else float(getattr(rob_t, "link_margin", 0.01))
        robot = RobotPointMass(int(need(rob_t, "q_dim")), margin)
    else:
        raise NotImplementedError(f"robot {rname!r}: the kernels know the point mass (2-D / 3-D) and the Panda")
    return PlanningTask(env, robot, obstacle_cutoff_margin=float(need(tr_task, "obstacle_cutoff_margin")), use_extra_objects=use_extra_objects,
                        tensor_args=tensor_args)


def compute_smoothness(trajs, robot, task=None):
    """sum_h |v_{h+1} - v_h| per trajectory (torch_robotics.trajectory.metrics.compute_smoothness, restated)."""
    v = robot.get_velocity(trajs)
    return torch.linalg.norm(torch.diff(v, dim=-2), dim=-1).sum(-1)


def compute_path_length(trajs, robot):
    q = robot.get_position(trajs)
    return torch.linalg.norm(torch.diff(q, dim=-2), dim=-1).sum(-1)


def compute_variance_waypoints(trajs, robot, definition="position_variance"):
    """Diversity of a batch of trajectories, summed over the waypoints (torch_robotics.trajectory.metrics.compute_variance_waypoints,
    un-vendored: restated; the definition is undecidable from the reference tree, so both plausible ones are offered):
      'position_variance' (default): sum_h sum_j Var_b[q_{b,h,j}]  (unbiased variance of each coordinate across the batch)
      'pairwise_distance'          : sum_h Var over unordered trajectory pairs of |q_{a,h} - q_{b,h}|  (SURVEY.md A20's recollection)"""
    q = robot.get_position(trajs)
    if q.shape[0] < 2:
        return 0.0
    if definition == "position_variance":
        return float(q.var(dim=0).sum(-1).sum())
    if definition == "pairwise_distance":
        B = q.shape[0]
        ia, ib = torch.triu_indices(B, B, offset=1, device=q.device)
        d = torch.linalg.norm(q[ia] - q[ib], dim=-1)     # [pairs, H]
        return float(d.var(dim=0).sum()) if d.shape[0] > 1 else 0.0
    raise ValueError(definition)


# ------------------------------------------------------------------------------------------------ cost descriptors


class CostCollision:
    def __init__(self, robot, n_support_points, field=None, sigma_coll=1.0, tensor_args=None, **kw):
        if sigma_coll != 1.0:
            raise NotImplementedError("sigma_coll != 1 (inference.py:201 always passes 1.0)")
        self.robot, self.n_support_points, self.field = robot, n_support_points, field


class CostGPTrajectory:
    """half_factor (extension, default False): whether the GP cost carries GPMP2's 1/2 (1/2 sum e^T Qinv e).  The reference's
    implementation lives in an empty submodule, so the convention is undecidable here; it only matters where the per-waypoint
    gradient norm is below max_grad_norm."""

    def __init__(self, robot, n_support_points, dt, sigma_gp=1.0, tensor_args=None, half_factor=False, **kw):
        self.robot, self.n_support_points, self.dt, self.sigma_gp = robot, n_support_points, float(dt), float(sigma_gp)
        self.half_factor = bool(half_factor)


class CostComposite:
    def __init__(self, robot, n_support_points, cost_list, weights_cost_l=None, tensor_args=None, **kw):
        self.robot, self.n_support_points = robot, n_support_points
        self.cost_l = list(cost_list)
        self.weight_cost_l = list(weights_cost_l) if weights_cost_l is not None else [1.0] * len(self.cost_l)
        if len(self.cost_l) != len(self.weight_cost_l):
            raise ValueError("one weight per cost term")
