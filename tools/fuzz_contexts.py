"""Randomised fuzz of the multi-context plan (BASELINE configs[4] in miniature; dev tool): C start/goal contexts x n trajectories in ONE plan (per-trajectory hard
conditions, per-context LimitsNormalizer range test) must equal one plan per context, bit for bit - with noise scaled so that some contexts leave the +-1 range
(the whole-tensor clip branch, normalization.py:156-167) and others do not.  python tools/fuzz_contexts.py [n_cases] [seed]"""
import random
import sys
from math import ceil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch   # noqa: E402
import mpd_public_amd as m   # noqa: E402
from mpd_public_amd.parallel import plan_contexts   # noqa: E402
from helpers import synth_sd, t, DIM_MULTS   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    env_id, robot_id = rng.choice([("EnvDense2D", "RobotPointMass"), ("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
    C, n = rng.choice([2, 3, 5, 9]), rng.choice([1, 2, 4, 7, 50])
    opt, T, n0 = rng.choice([0, 1]), 25, rng.choice([0, 5])
    n_guide, t_start = rng.choice([1, 5]), rng.choice([ceil(0.25 * T), T])
    desc = f"{env_id}-{robot_id} C={C} n={n} opt={opt} n0={n0} n_guide={n_guide} t_start={t_start}"
    try:
        ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
        D, q = ds.state_dim, ds.state_dim // 2
        net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
        net.load_state_dict(synth_sd(D, opt), strict=True)
        dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True).cuda().eval()
        H_, dt_ = 64, 5.0 / 64
        cl = [m.CostCollision(ds.robot, H_, field=f, sigma_coll=1.0) for f in ds.task.get_collision_fields()]
        wl = [1e-2] * len(cl)
        cl.append(m.CostGPTrajectory(ds.robot, H_, dt_, sigma_gp=1.0)); wl.append(1e-7)
        g = m.GuideManagerTrajectoriesWithVelocity(ds, m.CostComposite(ds.robot, H_, cl, weights_cost_l=wl), clip_grad=True, interpolate_trajectories_for_collision=True).cuda()
        noise = t(f"fc_noise/{case}", (T + n0 + 1, C * n, 64, D)).cuda()
        for c in range(C):   # every other context starts far outside the +-1 range, the rest inside
            noise[0, c * n:(c + 1) * n] *= 1.6 if c % 2 else 0.3
        mk = lambda tag, c: ds.normalizer.normalize(torch.cat([t(f"{tag}/{case}/{c}", (q,), "uniform", 0.7).cuda(), torch.zeros(q, device="cuda")]))
        starts, goals = torch.stack([mk("fcs", c) for c in range(C)]), torch.stack([mk("fcg", c) for c in range(C)])
        kw = dict(n_diffusion_steps_without_noise=n0, noise_std_extra_schedule_fn=lambda tt: 0.5, guide=g, n_guide_steps=n_guide, t_start_guide=t_start)
        batched, (lo, hi) = plan_contexts(dm, starts, goals, n, horizon=64, noise=noise, **kw)
        ok = (lo, hi) == (0, C) and bool(torch.isfinite(batched).all())
        for c in range(C):
            x, _ = dm.plan({0: starts[c], 63: goals[c]}, n, 64, noise=noise[:, c * n:(c + 1) * n].contiguous(), return_chain=False, **kw)
            ok = ok and bool(torch.equal(batched[c * n:(c + 1) * n], x))
        bad += 0 if ok else 1
        print(f"{'ok' if ok else 'MISMATCH'} case {case}: {desc}")
    except Exception as e:
        print(f"refused case {case}: {desc}: {type(e).__name__}: {str(e)[:150]}")
print(f"{n_cases} cases, {bad} mismatches")
