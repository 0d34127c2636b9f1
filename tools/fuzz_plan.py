"""Randomised configuration fuzz of the planning loop on one GPU (dev tool): for random (robot, batch, T, n0, guide steps, guide start, interpolation points,
dim_mults option, clip rule, scale_grad_by_std) the fused plan (ONE mpdx_plan enqueue) must equal the step-by-step protocol loop
(p_sample_loop -> ddpm_sample_fn -> guide_gradient_steps -> guide(x): one library call per piece) BIT FOR BIT - a size-independent property
that needs no oracle.  python tools/fuzz_plan.py [n_cases] [seed]"""
import random
import sys
from math import ceil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch   # noqa: E402
import mpd_public_amd as m   # noqa: E402
from helpers import synth_sd, t, DIM_MULTS   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    env_id, robot_id = rng.choice([("EnvDense2D", "RobotPointMass"), ("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSimple2D", "RobotPointMass"),
                                   ("EnvSpheres3D", "RobotPanda")])
    opt = rng.choice([0, 1])
    H = rng.choice([64, 64, 64, 32, 48, 96, 128])
    sched, T = rng.choice([("exponential", 25), ("exponential", 47), ("exponential", 50), ("cosine", 5), ("cosine", 12), ("cosine", 40)])   # (the reference's exponential schedule is finite for few T)
    n0 = rng.choice([0, 1, 5])
    B = rng.choice([1, 2, 3, 7, 16, 33, 100, 130])
    n_guide = rng.choice([0, 1, 3, 5])
    t_start = rng.choice([0, 1, ceil(0.25 * T), T, T + 3])
    n_interp = rng.choice([64, 96, 128]) if robot_id == "RobotPanda" else rng.choice([64, 96, 128, 200, 256])
    n_interp = max(n_interp, H)
    interp = rng.choice([True, True, False])
    clip_rule = rng.choice(["norm", "norm", "value"])
    sgs = rng.choice([False, False, True])
    guided = rng.random() < 0.8
    desc = (f"{env_id}-{robot_id} H={H} opt={opt} {sched} T={T} n0={n0} B={B} guided={guided} n_guide={n_guide} t_start={t_start} n_interp={n_interp} interp={interp} "
            f"clip={clip_rule} scale_by_std={sgs}")
    try:
        ds = m.TrajectoryDataset(env_id, robot_id, n_support_points=H, tensor_args={"device": "cuda", "dtype": torch.float32})
        D = ds.state_dim
        net = m.TemporalUnet(n_support_points=H, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
        net.load_state_dict(synth_sd(D, opt), strict=True)
        dm = m.GaussianDiffusionModel(model=net, variance_schedule=sched, n_diffusion_steps=T, predict_epsilon=True).cuda().eval()
        noise = t(f"fuzz_noise/{case}", (T + n0 + 1, B, H, D)).cuda()
        start = ds.normalizer.normalize(torch.cat([t(f"fs/{case}", (D // 2,), "uniform", 0.6).cuda(), torch.zeros(D // 2, device="cuda")]))
        goal = ds.normalizer.normalize(torch.cat([t(f"fg/{case}", (D // 2,), "uniform", 0.6).cuda(), torch.zeros(D // 2, device="cuda")]))
        hc = {0: start, H - 1: goal}
        kw = dict(n_samples=B, horizon=H, return_chain=True, sample_fn=m.ddpm_sample_fn, n_diffusion_steps_without_noise=n0,
                  noise_std_extra_schedule_fn=lambda tt: 0.5, noise=noise)
        if guided:
            H_, dt_ = H, 5.0 / H
            cl = [m.CostCollision(ds.robot, H_, field=f, sigma_coll=1.0) for f in ds.task.get_collision_fields()]
            wl = [1e-2] * len(cl)
            cl.append(m.CostGPTrajectory(ds.robot, H_, dt_, sigma_gp=1.0)); wl.append(1e-7)
            g = m.GuideManagerTrajectoriesWithVelocity(ds, m.CostComposite(ds.robot, H_, cl, weights_cost_l=wl), clip_grad=True, clip_grad_rule=clip_rule,
                                                       interpolate_trajectories_for_collision=interp, num_interpolated_points_for_collision=n_interp).cuda()
            kw.update(guide=g, n_guide_steps=n_guide, t_start_guide=t_start, scale_grad_by_std=sgs)
        a = dm.run_inference(None, hc, fused=True, **kw)
        b = dm.run_inference(None, hc, fused=False, **kw)
        ok = bool(torch.equal(a, b)) and bool(torch.isfinite(a).all())
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {desc}: max|fused - stepwise| {float((a - b).abs().max()):.3e} finite={bool(torch.isfinite(a).all())}")
        else:
            print(f"ok case {case}: {desc}")
    except Exception as e:   # a configuration the library refuses must say so loudly (RuntimeError / NotImplementedError / ValueError), never crash
        print(f"refused case {case}: {desc}: {type(e).__name__}: {str(e)[:140]}")
print(f"{n_cases} cases, {bad} mismatches")
