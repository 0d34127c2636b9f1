"""Post-loop trajectory metrics (oracle; TEST INFRASTRUCTURE - only tests/, smoke() and bench.py's cpu_baseline leg may import this).

Restates the three metrics scripts/inference/inference.py imports from the un-vendored torch_robotics
(`from torch_robotics.trajectory.metrics import compute_smoothness, compute_path_length, compute_variance_waypoints`,
inference.py:24) and applies to the collision-free trajectories (inference.py:311-327).  PARITY UNPINNED: the sources are an empty
git submodule, the reference holds no vectors for them; smoothness / path length follow SURVEY.md A20, the waypoint variance is
offered in the two definitions DESIGN.md section 5 names.  Written as plain float64 loops, independent of the product's tensor code.
"""
import numpy as np


def compute_path_length(trajs: np.ndarray, q_dim: int) -> np.ndarray:
    """sum_h |q_{h+1} - q_h|_2 per trajectory (inference.py:315)."""
    B, H = trajs.shape[:2]
    out = np.zeros(B)
    for b in range(B):
        for h in range(H - 1):
            d = trajs[b, h + 1, :q_dim].astype(np.float64) - trajs[b, h, :q_dim].astype(np.float64)
            out[b] += float(np.sqrt((d * d).sum()))
    return out


def compute_smoothness(trajs: np.ndarray, q_dim: int) -> np.ndarray:
    """sum_h |v_{h+1} - v_h|_2 per trajectory (inference.py:312)."""
    B, H = trajs.shape[:2]
    out = np.zeros(B)
    for b in range(B):
        for h in range(H - 1):
            d = trajs[b, h + 1, q_dim:2 * q_dim].astype(np.float64) - trajs[b, h, q_dim:2 * q_dim].astype(np.float64)
            out[b] += float(np.sqrt((d * d).sum()))
    return out


def compute_variance_waypoints(trajs: np.ndarray, q_dim: int, definition: str = "position_variance") -> float:
    """Diversity of a batch of trajectories summed over the waypoints (inference.py:327).
    'position_variance': sum_h sum_j Var_b[q_{b,h,j}] with the unbiased (n - 1) variance across the batch;
    'pairwise_distance': sum_h Var over the unordered trajectory pairs (a < b) of |q_{a,h} - q_{b,h}|_2 (unbiased)."""
    B, H = trajs.shape[:2]
    if B < 2:
        return 0.0
    q = trajs[..., :q_dim].astype(np.float64)
    total = 0.0
    for h in range(H):
        if definition == "position_variance":
            for j in range(q_dim):
                col = [q[b, h, j] for b in range(B)]
                mean = sum(col) / B
                total += sum((c - mean) ** 2 for c in col) / (B - 1)
        elif definition == "pairwise_distance":
            ds = []
            for a in range(B):
                for b in range(a + 1, B):
                    d = q[a, h] - q[b, h]
                    ds.append(float(np.sqrt((d * d).sum())))
            if len(ds) > 1:
                mean = sum(ds) / len(ds)
                total += sum((d - mean) ** 2 for d in ds) / (len(ds) - 1)
        else:
            raise ValueError(definition)
    return total
