"""Host logic of the training path that needs no GPU: the training-set loader (mpd/datasets/trajectories.py:84-172), the data
loaders of train.py, and that the native training step refuses to run without the GPU (no CPU fallback)."""
import numpy as np
import pytest
import torch


def _write_shards(base, sizes, D=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate(sizes):
        d = base / str(i)
        d.mkdir(parents=True)
        tr = torch.rand(n, 64, D, generator=g) * 1.6 - 0.8
        torch.save(tr, d / "trajs-free.pt")
        out.append(tr)
    (base / "empty").mkdir()
    return torch.cat(out)


def test_training_set_loader_matches_reference_semantics(tmp_path):
    from mpd_public_amd.datasets import TrajectoryDataset
    base = tmp_path / "EnvSimple2D-RobotPointMass"
    allt = _write_shards(base, [5, 3, 4])
    ds = TrajectoryDataset(env_id="EnvSimple2D", robot_id="RobotPointMass", base_dir=str(base))
    assert len(ds) == 12 and ds.n_support_points == 64 and ds.state_dim == 4
    assert sorted(ds.map_task_id_to_trajectories_id) == [0, 1, 2] and ds.map_trajectory_id_to_task_id[5] == 1
    flat = allt.reshape(-1, 4)
    assert torch.equal(ds.normalizer.mins, flat.min(0).values) and torch.equal(ds.normalizer.maxs, flat.max(0).values)   # normalization.py:144-153
    xn = ds.fields["traj_normalized"]
    assert float(xn.min()) == -1.0 and float(xn.max()) == 1.0
    assert torch.allclose(ds.unnormalize_trajectories(xn), ds.fields["traj"], atol=1e-6)
    item = ds[7]
    assert set(item) == {"traj_normalized", "task_normalized", "hard_conds"} and set(item["hard_conds"]) == {0, 63}
    # hard conditions: the trajectory's own end points with zero velocity (trajectories.py:205-223)
    assert torch.equal(item["hard_conds"][0][:2], xn[7, 0, :2]) and torch.equal(item["hard_conds"][0][2:], torch.zeros(2))
    assert ds.fields["task"].shape == (12, 4)


@pytest.mark.parametrize("name", ["GaussianNormalizer", "SafeLimitsNormalizer", "FixedLimitsNormalizer", "Identity"])
def test_training_set_with_the_other_normalizers(tmp_path, name):
    """TrajectoryDataset(normalizer=...) (trajectories.py:26,78): each field normalised by the named class built from the flattened field, as
    DatasetNormalizer does (normalization.py:14-22); the classes themselves are pinned to the reference in test_oracle_golden.py"""
    from mpd_public_amd.datasets import TrajectoryDataset
    from oracle.normalizer import from_data
    base = tmp_path / "EnvSimple2D-RobotPointMass"
    allt = _write_shards(base, [5, 3, 4])
    ds = TrajectoryDataset(env_id="EnvSimple2D", robot_id="RobotPointMass", base_dir=str(base), normalizer=name)
    assert type(ds.normalizer).__name__ == name
    want = from_data(name, allt.reshape(-1, 4))
    assert torch.equal(ds.fields["traj_normalized"], want.normalize(allt))
    assert torch.allclose(ds.unnormalize_trajectories(ds.fields["traj_normalized"]), allt, atol=1e-6)
    task = torch.cat((allt[:, 0, :2], allt[:, -1, :2]), -1)
    assert torch.equal(ds.fields["task_normalized"], from_data(name, task).normalize(task))
    with pytest.raises(NameError):
        TrajectoryDataset(env_id="EnvSimple2D", robot_id="RobotPointMass", normalizer="NoSuchNormalizer")
    with pytest.raises(ValueError):
        TrajectoryDataset(env_id="EnvSimple2D", robot_id="RobotPointMass", normalizer="GaussianNormalizer")   # needs data


def test_get_dataset_split_and_batches(tmp_path):
    from mpd_public_amd import train as train_script
    base = tmp_path / "data" / "EnvSimple2D-RobotPointMass"
    _write_shards(base, [20, 20])
    tr, trl, va, val = train_script.get_dataset(dataset_subdir="EnvSimple2D-RobotPointMass", batch_size=8, val_set_size=0.1,
                                               results_dir=str(tmp_path), save_indices=True, data_dir=str(tmp_path / "data"))
    assert len(tr) == 36 and len(va) == 4 and (tmp_path / "train_subset_indices.pt").exists()
    b = next(iter(trl))
    assert b["traj_normalized"].shape == (8, 64, 4) and b["hard_conds"][63].shape == (8, 4)
    from mpd_public_amd.trainer import get_num_epochs
    assert get_num_epochs(100, 8, 36) == 23   # trainer.py:16-17


def test_batch_gather_loader_yields_the_dataloaders_batches(tmp_path):
    """train.BatchGatherLoader (one gather per batch) against torch's DataLoader over the same Subset - the reference's loaders
    (train_loaders.py:92-93: sequential sampler, default collate): every batch, the ragged last one included, bit for bit."""
    from torch.utils.data import DataLoader
    from mpd_public_amd import train as train_script
    base = tmp_path / "data" / "EnvSimple2D-RobotPointMass"
    _write_shards(base, [13, 9, 7])
    tr, trl, va, val = train_script.get_dataset(dataset_subdir="EnvSimple2D-RobotPointMass", batch_size=8, val_set_size=0.1,
                                               data_dir=str(tmp_path / "data"))
    assert isinstance(trl, train_script.BatchGatherLoader) and isinstance(val, train_script.BatchGatherLoader)
    ref = DataLoader(tr, batch_size=8)
    assert len(trl) == len(ref) == 4
    n = 0
    for a, b in zip(trl, ref):
        assert set(a) == set(b) and set(a["hard_conds"]) == set(b["hard_conds"]) == {0, 63}
        assert torch.equal(a["traj_normalized"], b["traj_normalized"]) and torch.equal(a["task_normalized"], b["task_normalized"])
        for k in (0, 63):
            assert torch.equal(a["hard_conds"][k], b["hard_conds"][k])
        n += len(a["traj_normalized"])
    assert n == len(tr) == 26
    assert [len(b["traj_normalized"]) for b in trl] == [8, 8, 8, 2]   # (a second pass: the loader is re-iterable, like a DataLoader)


def test_training_step_has_no_cpu_fallback():
    import mpd_public_amd as m
    from mpd_public_amd.trainer import TrainStep
    net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4))
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True)
    with pytest.raises(RuntimeError, match="GPU"):
        TrainStep(dm)


def test_early_stopper_mirrors_the_reference():
    """trainer.py:45-64: stops after `patience` validations above the best; patience -1 never stops."""
    from mpd_public_amd.trainer import EarlyStopper
    e = EarlyStopper(patience=2)
    assert [e.early_stop(v) for v in (1.0, 0.9, 0.95, 0.96)] == [False, False, False, True]
    assert not any(EarlyStopper(patience=-1).early_stop(v) for v in (1.0, 2.0, 3.0, 4.0))
