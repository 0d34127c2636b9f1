"""Dev tool: one case of tools/fuzz_planner.py looked at trajectory by trajectory - is a loss of collision-free trajectories under GPMP2 the ALGORITHM's
(the float64 oracle, run with the kernel's accept / reject rule on the same initial trajectory, loses them too) or the kernel's?
python tools/gpmp_case_probe.py <fuzz seed> <case index> [n trajectories to replay on the oracle]"""
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch   # noqa: E402
import mpd_public_amd as m   # noqa: E402
from mpd_public_amd.generate_trajectories import GPMP2, RRTConnectBatch   # noqa: E402
from helpers import oracle_guide   # noqa: E402
from oracle import gpmp as ogpmp   # noqa: E402

fseed, want = int(sys.argv[1]), int(sys.argv[2])
n_replay = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = random.Random(fseed)
for case in range(want + 1):   # the draws of fuzz_planner.py, in its order
    env_id, robot_id = rng.choice([("EnvSimple2D", "RobotPointMass"), ("EnvDense2D", "RobotPointMass"), ("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")])
    n = rng.choice([1, 3, 8, 16, 40])
    seed = rng.randrange(1000)
print(f"case {want}: {env_id}-{robot_id} n={n} context seed={seed}")
ds = m.TrajectoryDataset(env_id, robot_id, n_support_points=64, obstacle_cutoff_margin=0.03, tensor_args={"device": torch.device("cuda"), "dtype": torch.float32})
gen = torch.Generator(device="cuda").manual_seed(seed)
for _ in range(200):
    q = ds.task.random_coll_free_q(n_samples=2, device="cuda", generator=gen)
    if torch.linalg.norm(q[0] - q[1]) > ds.threshold_start_goal_pos:
        break
dt = 5.0 / 64
rrt = RRTConnectBatch(ds.task, q[0], q[1], n, step_size=0.1 if ds.robot.q_dim <= 3 else 0.25, generator=gen)
rrt.grow(max_iters=6000)
x0 = rrt.trajectories(64, dt)
opt = GPMP2(ds, dt, device="cuda")
x = x0
for k in range(4):
    x = opt.optimize(x, opt_iters=60 if k else 1)
hits0, hits1 = ds.task.trajectory_metrics(x0)[:, 0].cpu(), ds.task.trajectory_metrics(x)[:, 0].cpu()
print("solved by RRT-Connect:", rrt.done.cpu().int().tolist())
print("colliding waypoints before:", hits0.int().tolist())
print("colliding waypoints after: ", hits1.int().tolist())
_, comp = oracle_guide(ds, 1.0, 1.0, clip_grad=False, dtype=torch.float64)
coll = comp.cost_l[:-1]
for c in coll:
    c.cutoff = ds.task.obstacle_cutoff_margin
robot, o = coll[0].robot, opt.opts
lost = [i for i in range(n) if hits0[i] == 0 and hits1[i] > 0][:n_replay]
for i in lost:
    th = x0[i].cpu().double()
    args = (robot, coll, dt, 1.0, o.sigma_obs, 128)
    F, lam, acc = float(ogpmp.objective(th, *args)), opt.lambda_init, 0
    for it in range(181):   # the kernel's rule (planner.hpp: accept a lower objective and relax lambda, else stiffen it)
        d, _ = ogpmp.lm_step(th, *args, lam)
        Fc = float(ogpmp.objective(th + d, *args))
        if Fc < F:
            th, F, lam, acc = th + d, Fc, max(lam * o.lambda_down, o.lambda_min), acc + 1
        else:
            lam = min(lam * o.lambda_up, o.lambda_max)
    h = int(ds.task.trajectory_metrics(th.float().cuda()[None])[0, 0])
    print(f"trajectory {i}: kernel F {float(opt.state[i, 0]):.4f}, {int(hits1[i])} colliding waypoints | float64 oracle replay F {F:.4f}, {h} colliding waypoints, "
          f"{acc} accepted steps, max |x_kernel - x_oracle| {float((x[i].cpu().double() - th).abs().max()):.3e}")
