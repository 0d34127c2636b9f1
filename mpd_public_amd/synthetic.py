"""Deterministic synthetic inputs (weights, normaliser limits, obstacle sets, start/goal pairs).

Nothing real is available for this path: the reference ships no trained weights, no dataset and
no environment geometry (reference README.md:69-72 points at Google-Drive tarballs; deps/ is empty).
SURVEY.md section 8(d) therefore fixes *formula-defined* inputs that every box can re-create
bit-identically.  Everything here is integer-hash based (splitmix64 over the flat element index),
so it does not depend on numpy's Generator stream stability, libm, or the platform.

Used by: tests/golden/make_golden.py (to load the *reference* modules with these weights),
the parity tests, bench.py and __graft_entry__.smoke().
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _name_seed(name: str) -> np.uint64:
    """FNV-1a 64-bit over the utf-8 bytes of ``name`` (stable across runs and platforms)."""
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, n: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """``n`` float64 values in [lo, hi) that depend only on (name, flat index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(idx ^ _name_seed(name))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)  # 53-bit mantissa
    return lo + (hi - lo) * u


def hash_normal(name: str, n: int) -> np.ndarray:
    """Standard normals from two hash-uniform streams (Box-Muller, float64)."""
    u1 = hash_uniform(name + "/u1", n, 0.0, 1.0)
    u2 = hash_uniform(name + "/u2", n, 0.0, 1.0)
    u1 = np.maximum(u1, 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def synth_param(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """Closed-form fp32 value for a U-Net parameter, keyed by its state-dict name.

    * conv / linear ``weight``: U(-b, b) with b = 1/sqrt(fan_in) (the scale torch's default init uses,
      so activations stay O(1) through the 33 conv blocks);
    * GroupNorm ``block.2.weight``: 1 + 0.1 u,  ``block.2.bias``: 0.1 u;
    * other ``bias``: U(-b, b) with b = 1/sqrt(fan_out-ish) -> use 0.05 so biases matter but do not dominate.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n)
    if name.endswith("block.2.weight"):
        v = 1.0 + 0.1 * u
    elif name.endswith("block.2.bias"):
        v = 0.1 * u
    elif name.endswith("weight"):
        if len(shape) == 3:  # Conv1d [Co,Ci,k] / ConvTranspose1d [Ci,Co,k]: fan_in = shape[1]*k either way is fine
            fan_in = shape[1] * shape[2]
        elif len(shape) == 2:
            fan_in = shape[1]
        else:
            fan_in = max(n, 1)
        v = u / math.sqrt(fan_in)
    else:
        v = 0.05 * u
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]]):
    """name -> torch fp32 tensor for every (name, shape)."""
    import torch

    return {k: torch.from_numpy(synth_param(k, tuple(s)).copy()) for k, s in shapes.items()}


def synth_tensor(name: str, shape: Iterable[int], kind: str = "normal", scale: float = 1.0) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    v = hash_normal(name, n) if kind == "normal" else hash_uniform(name, n)
    return (scale * v).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------------------
# normaliser limits, environments (SURVEY.md section 8(d) "Synthetic inputs")
# ----------------------------------------------------------------------------------------------

PANDA_Q_MIN = np.array([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973], dtype=np.float32)
PANDA_Q_MAX = np.array([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973], dtype=np.float32)


def limits_for(robot: str) -> Tuple[np.ndarray, np.ndarray]:
    """(mins[D], maxs[D]) of the trajectory normaliser; D = 2*q_dim (positions then velocities)."""
    if robot == "RobotPointMass":
        q_min, q_max, v = np.full(2, -1.0, np.float32), np.full(2, 1.0, np.float32), 2.0
    elif robot == "RobotPointMass3D":
        q_min, q_max, v = np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32), 2.0
    elif robot == "RobotPanda":
        q_min, q_max, v = PANDA_Q_MIN, PANDA_Q_MAX, 2.5
    else:
        raise NotImplementedError(robot)
    vv = np.full_like(q_min, v)
    return np.concatenate([q_min, -vv]), np.concatenate([q_max, vv])
