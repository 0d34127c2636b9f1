"""`experiment(...)` of scripts/train_diffusion/train.py:16-150 with the same arguments: dataset -> TemporalUnet +
GaussianDiffusionModel -> trainer.train (native HIP training step).  Writes what inference.py reads back:
`<results_dir>/args.yaml`, `checkpoints/{model,ema_model}_current_state_dict.pth` and `limits.yaml` (the normaliser limits of
the training set, which the reference re-derives by re-loading the dataset at inference time).

    python -m mpd_public_amd.train --dataset_subdir EnvSimple2D-RobotPointMass --data_dir data_trajectories --num_train_steps 2000
"""
from __future__ import annotations

import argparse
import os

import torch
import yaml

from . import trainer
from . import summaries
from .datasets import TrajectoryDataset
from .diffusion_model import GaussianDiffusionModel
from .temporal_unet import TemporalUnet, UNET_DIM_MULTS


class BatchGatherLoader:
    """What `DataLoader(subset, batch_size=batch_size)` yields for a TrajectoryDataset (sequential sampler, default collate: the reference's loaders,
    train_loaders.py:92-93) - the same batches, bit for bit, each built with ONE gather of the dataset's tensors instead of `batch_size` Python
    `__getitem__` calls and a collate: the per-sample path costs ~1 ms of host time per batch of 32, more than the native training step takes
    (0.6 ms), so the GPU idled for 60 % of a training run.  Iteration protocol and `len()` as the DataLoader's."""

    def __init__(self, subset, batch_size, drop_last=False):
        self.dataset, self.batch_size, self.drop_last = subset, int(batch_size), drop_last
        base = subset.dataset if hasattr(subset, "indices") else subset
        idx = subset.indices if hasattr(subset, "indices") else range(len(subset))
        self._base = base
        self._idx = torch.as_tensor(list(idx), dtype=torch.long, device=base.fields["traj_normalized"].device)
        # TrajectoryDataset.get_hard_conditions of EVERY trajectory, once (trajectories.py:205-223: end positions, zero velocities): a batch is
        # then four gathers - no slicing / cat / zeros launches per batch
        x = base.fields["traj_normalized"]
        s, g = base.robot.get_position(x[:, 0]), base.robot.get_position(x[:, -1])
        if base.include_velocity:
            s, g = torch.cat((s, torch.zeros_like(s)), dim=-1), torch.cat((g, torch.zeros_like(g)), dim=-1)
        self._hs, self._hg = s.contiguous(), g.contiguous()

    def __len__(self):
        n = len(self._idx)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        b, n = self._base, len(self._idx)
        H = b.fields["traj_normalized"].shape[1]
        for i in range(0, n, self.batch_size):
            idx = self._idx[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                return
            yield {"traj_normalized": b.fields["traj_normalized"].index_select(0, idx), "task_normalized": b.fields["task_normalized"].index_select(0, idx),
                   "hard_conds": {0: self._hs.index_select(0, idx), H - 1: self._hg.index_select(0, idx)}}


def get_dataset(dataset_class="TrajectoryDataset", dataset_subdir=None, batch_size=2, val_set_size=0.05, results_dir=None,
                save_indices=False, data_dir="data_trajectories", tensor_args=None, seed=0, **kwargs):
    """mpd/trainer/train_loaders.py:77-99: full dataset, random split, two DataLoaders (batches stay on the dataset's device)."""
    from torch.utils.data import DataLoader, random_split
    env_id, robot_id = dataset_subdir.split("-")[0], dataset_subdir.split("-")[1]
    full = TrajectoryDataset(env_id=env_id, robot_id=robot_id, base_dir=os.path.join(data_dir, dataset_subdir), tensor_args=tensor_args, **kwargs)
    n_val = max(1, int(round(len(full) * val_set_size)))
    gen = torch.Generator().manual_seed(seed)
    train_subset, val_subset = random_split(full, [len(full) - n_val, n_val], generator=gen)
    # the reference's loaders (sequential, default collate); BatchGatherLoader yields the identical batches without the per-sample Python path
    # (MPDX_TORCH_DATALOADER=1 keeps torch's DataLoader)
    if os.environ.get("MPDX_TORCH_DATALOADER", "0") == "1":
        train_dataloader = DataLoader(train_subset, batch_size=batch_size)
        val_dataloader = DataLoader(val_subset, batch_size=batch_size)
    else:
        train_dataloader = BatchGatherLoader(train_subset, batch_size)
        val_dataloader = BatchGatherLoader(val_subset, batch_size)
    if save_indices and results_dir is not None:
        torch.save(train_subset.indices, os.path.join(results_dir, "train_subset_indices.pt"))
        torch.save(val_subset.indices, os.path.join(results_dir, "val_subset_indices.pt"))
    return train_subset, train_dataloader, val_subset, val_dataloader


def experiment(dataset_subdir: str = "EnvSimple2D-RobotPointMass", include_velocity: bool = True,
               diffusion_model_class: str = "GaussianDiffusionModel", variance_schedule: str = "exponential", n_diffusion_steps: int = 25,
               predict_epsilon: bool = True, unet_input_dim: int = 32, unet_dim_mults_option: int = 1,
               loss_class: str = "GaussianDiffusionLoss", batch_size: int = 32, lr: float = 1e-4, num_train_steps: int = 500000,
               use_ema: bool = True, use_amp: bool = False, steps_til_summary: int = 10, summary_class: str = "SummaryTrajectoryGeneration",
               steps_til_ckpt: int = 50000, device: str = "cuda", debug: bool = True, seed: int = 0, results_dir: str = "logs",
               data_dir: str = "data_trajectories", **kwargs):
    if diffusion_model_class != "GaussianDiffusionModel" or loss_class != "GaussianDiffusionLoss":
        raise NotImplementedError("only GaussianDiffusionModel / GaussianDiffusionLoss (the classes train.py's defaults name)")
    torch.manual_seed(seed)
    os.makedirs(results_dir, exist_ok=True)
    tensor_args = {"device": device, "dtype": torch.float32}
    train_subset, train_dataloader, val_subset, val_dataloader = get_dataset(
        dataset_class="TrajectoryDataset", include_velocity=include_velocity, dataset_subdir=dataset_subdir, batch_size=batch_size,
        results_dir=results_dir, save_indices=True, data_dir=data_dir, tensor_args=tensor_args, seed=seed)
    dataset = train_subset.dataset
    unet_configs = dict(state_dim=dataset.state_dim, n_support_points=dataset.n_support_points, unet_input_dim=unet_input_dim,
                        dim_mults=UNET_DIM_MULTS[unet_dim_mults_option])
    model = GaussianDiffusionModel(model=TemporalUnet(**unet_configs), variance_schedule=variance_schedule,
                                   n_diffusion_steps=n_diffusion_steps, predict_epsilon=predict_epsilon).to(device)
    model.manual_seed(seed)
    args = dict(dataset_subdir=dataset_subdir, include_velocity=include_velocity, diffusion_model_class=diffusion_model_class,
                variance_schedule=variance_schedule, n_diffusion_steps=n_diffusion_steps, predict_epsilon=predict_epsilon,
                unet_input_dim=unet_input_dim, unet_dim_mults_option=unet_dim_mults_option, loss_class=loss_class, batch_size=batch_size,
                lr=lr, num_train_steps=num_train_steps, use_ema=use_ema, seed=seed)
    with open(os.path.join(results_dir, "args.yaml"), "w") as f:
        yaml.safe_dump(args, f)
    with open(os.path.join(results_dir, "limits.yaml"), "w") as f:
        nrm = dataset.normalizer
        lim = {"normalizer": type(nrm).__name__, "mins": [float(v) for v in nrm.mins.cpu()], "maxs": [float(v) for v in nrm.maxs.cpu()]}
        if hasattr(nrm, "means"):
            lim.update(means=[float(v) for v in nrm.means.cpu()], stds=[float(v) for v in nrm.stds.cpu()])
        yaml.safe_dump(lim, f)
    loss_fn = trainer.GaussianDiffusionLoss.loss_fn
    summary_fn = getattr(summaries, summary_class)(seed=seed).summary_fn if summary_class else None   # train_loaders.py:102-107
    model, ema_model, losses = trainer.train(
        model=model, train_dataloader=train_dataloader, train_subset=train_subset, val_dataloader=val_dataloader, val_subset=train_subset,
        epochs=trainer.get_num_epochs(num_train_steps, batch_size, len(dataset)), model_dir=results_dir, summary_fn=summary_fn, lr=lr,
        loss_fn=loss_fn, val_loss_fn=loss_fn, steps_til_summary=steps_til_summary, steps_til_checkpoint=steps_til_ckpt, clip_grad=True,
        use_ema=use_ema, use_amp=use_amp, debug=debug, tensor_args=tensor_args, max_steps=num_train_steps)
    return model, ema_model, losses


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default in (("dataset_subdir", str, "EnvSimple2D-RobotPointMass"), ("data_dir", str, "data_trajectories"),
                               ("results_dir", str, "logs"), ("n_diffusion_steps", int, 25), ("unet_dim_mults_option", int, 1),
                               ("batch_size", int, 32), ("lr", float, 1e-4), ("num_train_steps", int, 500000), ("steps_til_summary", int, 10),
                               ("steps_til_ckpt", int, 50000), ("seed", int, 0)):
        ap.add_argument(f"--{name}", type=typ, default=default)
    ap.add_argument("--summary_class", default="SummaryTrajectoryGeneration",   # train.py:48; "none" switches the summaries off
                    type=lambda v: None if v.lower() in ("none", "") else v)
    a = ap.parse_args(argv)
    experiment(**vars(a))


if __name__ == "__main__":
    main()
