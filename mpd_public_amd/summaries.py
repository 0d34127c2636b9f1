"""SummaryTrajectoryGeneration - the statistics half of mpd/summaries/summary_trajectory_generation.py:11-48: during training,
sample 25 trajectories for a random task of the data subset with the (EMA) model and report the fraction of collision-free
trajectories, the collision intensity and the success flag.  The reference logs them to wandb and renders figures (matplotlib);
neither is on the compute path nor installed here: the numbers are printed and appended to `self.history`."""
from __future__ import annotations

import numpy as np
import torch


class SummaryTrajectoryGeneration:
    def __init__(self, n_samples: int = 25, seed: int = 0, **kwargs):
        self.n_samples = n_samples
        self.rng = np.random.default_rng(seed)
        self.history = []

    def summary_fn(self, train_step=None, model=None, datasubset=None, prefix="", debug=False, **kwargs):
        dataset = datasubset.dataset
        trajectory_id = int(self.rng.choice(datasubset.indices))                      # :22
        data_normalized = dataset[trajectory_id]
        hard_conds = {k: v.to(next(model.parameters()).device) for k, v in data_normalized["hard_conds"].items()}
        with torch.no_grad():
            trajs_normalized = model.run_inference(None, hard_conds, n_samples=self.n_samples, horizon=dataset.n_support_points)   # :34-38
        trajs = dataset.unnormalize_trajectories(trajs_normalized)                    # :41
        task = dataset.task
        rec = {"train_step": train_step, "prefix": prefix,
               "percentage free trajs": float(task.compute_fraction_free_trajs(trajs)),            # :45
               "percentage collision intensity": float(task.compute_collision_intensity_trajs(trajs)),   # :46
               "success": float(task.compute_success_free_trajs(trajs))}                          # :47
        self.history.append(rec)
        print(f"{prefix}step {train_step}: free {rec['percentage free trajs']:.3f}  collision intensity "
              f"{rec['percentage collision intensity']:.4f}  success {rec['success']:.0f}")
        return rec
