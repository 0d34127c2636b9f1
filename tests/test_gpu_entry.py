"""GPU tests of the post-loop metrics kernel and of the experiment() entry."""
import numpy as np
import pytest
import torch

from helpers import obstacle_hugging_trajs, oracle_guide

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,robot_id", [("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
def test_trajectory_metrics_vs_oracle(env_id, robot_id):
    import mpd_public_amd as m
    from oracle import costs as oc
    from oracle.guide import interpolate_points_v1
    ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
    xn = obstacle_hugging_trajs(ds, 9, seed=f"met/{env_id}", scale=0.97)
    og, comp = oracle_guide(ds, dtype=torch.float64)
    xu = og.normalizer.unnormalize(xn.double())
    got = ds.task.trajectory_metrics(xu.float().cuda(), n_check=256).cpu().numpy()
    qd = ds.state_dim // 2
    q, v = xu[..., :qd], xu[..., qd:]
    np.testing.assert_allclose(got[:, 1], torch.linalg.norm(q[:, 1:] - q[:, :-1], dim=-1).sum(-1).numpy(), rtol=2e-6)
    np.testing.assert_allclose(got[:, 2], torch.linalg.norm(v[:, 1:] - v[:, :-1], dim=-1).sum(-1).numpy(), rtol=2e-6)
    # collision count: a waypoint collides iff some hinge with margin = link radius (no cutoff) is active
    xi = interpolate_points_v1(xu, 256)
    hit = torch.zeros(xi.shape[:2], dtype=torch.bool)
    for term in comp.cost_l:
        if isinstance(term, oc.CostCollision):
            term.cutoff = 0.0
            per_point = torch.stack([term(xi[:, i:i + 1]) for i in range(xi.shape[1])], 1)
            hit |= per_point > 0
    want = hit.sum(1).numpy()
    assert (got[:, 3] == 256).all()
    assert np.abs(got[:, 0] - want).max() <= 2, (got[:, 0], want)   # points within fp32 rounding of a surface may differ
    assert want.max() > 0, "the probe trajectories must contain collisions"
    # plan-level figures the entry reports (inference.py:293-297), HIP kernel vs oracle counts
    frac_free_got, frac_free_want = float((got[:, 0] == 0).mean()), float((want == 0).mean())
    inten_got, inten_want = float((got[:, 0] / got[:, 3]).mean()), float((want / 256.0).mean())
    assert frac_free_got == frac_free_want, (frac_free_got, frac_free_want)
    assert abs(inten_got - inten_want) <= 5e-3 * inten_want, (inten_got, inten_want)   # 3 significant figures


@pytest.mark.parametrize("model_id,planner", [("EnvDense2D-RobotPointMass", "mpd"), ("EnvSpheres3D-RobotPanda", "mpd"),
                                              ("EnvSimple2D-RobotPointMass", "diffusion_prior_then_guide"),
                                              ("EnvNarrowPassageDense2D-RobotPointMass", "diffusion_prior")])
def test_experiment_entry_runs_and_reports(tmp_path, model_id, planner):
    from mpd_public_amd.inference import experiment
    n = 12
    r = experiment(model_id=model_id, planner_alg=planner, n_samples=n, debug=False, results_dir=str(tmp_path), seed=3)
    T = 25
    steps = T + 5 + 1 + ((7 + 5) * 5 if planner == "diffusion_prior_then_guide" else 0)
    D = 14 if "Panda" in model_id else 4
    assert tuple(r["trajs_iters"].shape) == (steps, n, 64, D)
    assert torch.isfinite(r["trajs_iters"]).all()
    assert 0.0 <= r["fraction_free_trajs"] <= 1.0 and 0.0 <= r["collision_intensity_trajs"] <= 1.0
    assert r["t_total"] > 0
    nf = 0 if r["trajs_final_free"] is None else r["trajs_final_free"].shape[0]
    nc = 0 if r["trajs_final_coll"] is None else r["trajs_final_coll"].shape[0]
    assert nf + nc == n
    assert (tmp_path / model_id / "results_inference" / "3" / "results_data_dict.pickle").exists()
    # start/goal hard conditions hold on every element of the chain
    xs = r["trajs_iters"]
    assert torch.equal(xs[:, :, 0, :], xs[0:1, 0:1, 0, :].expand(steps, n, -1))
    assert torch.equal(xs[:, :, -1, :], xs[0:1, 0:1, -1, :].expand(steps, n, -1))


def test_experiment_loads_reference_format_checkpoint(tmp_path):
    """model_dir with args.yaml + checkpoints/ema_model_current_state_dict.pth (the layout trainer.py:29-37 writes)."""
    import yaml
    import mpd_public_amd as m
    from mpd_public_amd import synthetic as syn
    from mpd_public_amd.inference import experiment
    md = tmp_path / "EnvSimple2D-RobotPointMass"
    (md / "checkpoints").mkdir(parents=True)
    (md / "args.yaml").write_text(yaml.safe_dump(dict(variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True,
                                                      unet_input_dim=32, unet_dim_mults_option=0, use_ema=True, include_velocity=True)))
    dm = m.GaussianDiffusionModel(model=m.TemporalUnet(n_support_points=64, state_dim=4, dim_mults=(1, 2, 4)), n_diffusion_steps=25,
                                  predict_epsilon=True)
    sd = dm.state_dict()
    for k in sd:  # weights that differ from the synthetic default, so that loading is observable
        if k.startswith("model."):
            sd[k] = torch.from_numpy(syn.synth_param("ckpt/" + k, tuple(sd[k].shape)))
    torch.save(sd, md / "checkpoints" / "ema_model_current_state_dict.pth")
    # trained weights without the training set's normaliser limits are refused (the package's limits are synthetic) ...
    with pytest.raises(RuntimeError, match="limits.yaml"):
        experiment(model_dir=str(md), model_id="EnvSimple2D-RobotPointMass", n_samples=2, debug=False, results_dir=None, planner_alg="diffusion_prior")
    # ... unless they travel with the checkpoint
    (md / "limits.yaml").write_text(yaml.safe_dump(dict(mins=[-1.0, -1.0, -2.0, -2.0], maxs=[1.0, 1.0, 2.0, 2.0])))
    kw = dict(model_id="EnvSimple2D-RobotPointMass", n_samples=6, debug=False, results_dir=str(tmp_path / "out"), seed=5,
              planner_alg="diffusion_prior")
    a = experiment(model_dir=str(md), **kw)
    b = experiment(model_dir=None, model_args=dict(unet_dim_mults_option=0), **kw)
    assert a["trajs_iters"].shape == b["trajs_iters"].shape == (31, 6, 64, 4)
    assert torch.isfinite(a["trajs_iters"]).all()
    assert not torch.allclose(a["trajs_iters"][-1], b["trajs_iters"][-1])   # the checkpoint's weights were used


@pytest.mark.parametrize("robot_id,env_id", [("RobotPointMass", "EnvDense2D"), ("RobotPanda", "EnvSpheres3D")])
def test_plan_diversity_and_cost_metrics_vs_oracle(robot_id, env_id):
    """compute_variance_waypoints (both definitions), compute_smoothness, compute_path_length (inference.py:24,311-327; un-vendored:
    parity unpinned) on device tensors against the plain-loop float64 restatement of oracle/metrics.py."""
    import mpd_public_amd as m
    from mpd_public_amd.planning import compute_variance_waypoints, compute_smoothness, compute_path_length
    from oracle import metrics as om
    ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
    qd = ds.state_dim // 2
    xn = obstacle_hugging_trajs(ds, 9, seed=f"var/{env_id}", scale=0.9)
    xu = ds.unnormalize_trajectories(xn.cuda())
    ref = xu.cpu().numpy()
    np.testing.assert_allclose(compute_path_length(xu, ds.robot).cpu().numpy(), om.compute_path_length(ref, qd), rtol=5e-6)
    np.testing.assert_allclose(compute_smoothness(xu, ds.robot).cpu().numpy(), om.compute_smoothness(ref, qd), rtol=5e-6)
    for definition in ("position_variance", "pairwise_distance"):
        got = compute_variance_waypoints(xu, ds.robot, definition=definition)
        want = om.compute_variance_waypoints(ref, qd, definition)
        assert want > 0
        assert abs(got - want) <= 2e-5 * want, (definition, got, want)
    assert compute_variance_waypoints(xu[:1], ds.robot) == 0.0
    # best-trajectory selection of the entry (inference.py:319-322) on the same numbers
    cost = om.compute_path_length(ref, qd) + om.compute_smoothness(ref, qd)
    assert int(torch.argmin(compute_path_length(xu, ds.robot) + compute_smoothness(xu, ds.robot))) == int(np.argmin(cost))


def test_measurement_helpers_guide_time_and_unit_bytes():
    """bench.py's helpers behind the C ABI: mpdx_guide_time (gradient-only launches: x untouched, a positive time) and mpdx_unet_unit_bytes
    (algorithmic bytes of a pass's launch units: every weight once + boundary activations)."""
    import ctypes as C
    import mpd_public_amd as m
    from mpd_public_amd import _lib, synthetic as syn
    from helpers import product_guide
    lib = _lib.load()
    ds = m.TrajectoryDataset("EnvDense2D", "RobotPointMass", tensor_args={"device": "cuda", "dtype": torch.float32})
    g = product_guide(ds).cuda()
    B, H, D = 16, 64, ds.state_dim
    x = obstacle_hugging_trajs(ds, B, seed="gt", scale=0.9).cuda().contiguous()
    x0 = x.clone()
    gp = g.device_params(x.device)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.mpdx_absmax(x.data_ptr(), flag.data_ptr(), B, B, H, D, _lib.current_stream()))
    out, ms = torch.empty_like(x), C.c_float()
    _lib.check(lib.mpdx_guide_time(C.byref(gp), x.data_ptr(), out.data_ptr(), flag.data_ptr(), B, B, H, D, 20, _lib.current_stream(), C.byref(ms)))
    assert ms.value > 0 and torch.equal(x, x0)
    assert torch.equal(out, g(x))   # the timed launches computed the guide increment
    net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4, 8))
    net.load_state_dict(syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
    net = net.cuda()
    hdl = net.engine(100, B)[0]
    tot, i = 0.0, 0
    while lib.mpdx_unet_unit_layer(hdl, B, i) >= 0 or lib.mpdx_unet_unit_bytes(hdl, B, i) > 0:
        b = lib.mpdx_unet_unit_bytes(hdl, B, i)
        assert b > 0
        tot += b
        i += 1
    n_conv = sum(v.numel() for k, v in net.state_dict().items() if k.endswith("weight") and v.dim() == 3 and not k.startswith("final_conv.1"))
    assert i == 15 and tot > 4.0 * n_conv   # two programs + thirteen per-layer / paired launches; at least every conv weight once
    assert lib.mpdx_unet_unit_bytes(hdl, B, i) == 0.0
