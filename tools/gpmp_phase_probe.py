"""dev probe: where a GPMP2 Levenberg-Marquardt iteration (csrc/planner.hpp gpmp_lm_kernel) spends its time - launches with solve = 0 (judge +
linearise only) against full iterations - and what the narrow-passage success rate does with the obstacle factor's stiffness."""
import ctypes as C, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import mpd_public_amd as m
from mpd_public_amd import _lib
from mpd_public_amd.generate_trajectories import GPMP2, RRTConnectBatch
lib = _lib.load()
for env_id, robot_id in (("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")):
    ds = m.TrajectoryDataset(env_id, robot_id, n_support_points=64, obstacle_cutoff_margin=0.03, tensor_args={"device": torch.device("cuda"), "dtype": torch.float32})
    gen = torch.Generator(device="cuda").manual_seed(2)
    q = None
    for _ in range(100):
        q = ds.task.random_coll_free_q(n_samples=2, device="cuda", generator=gen)
        if torch.linalg.norm(q[0] - q[1]) > 1.0:
            break
    n, dt = 100, 5.0 / 64
    rrt = RRTConnectBatch(ds.task, q[0], q[1], n, step_size=0.1 if ds.robot.q_dim <= 3 else 0.25, generator=gen)
    rrt.grow(max_iters=6000)
    x0 = rrt.trajectories(64, dt)
    opt = GPMP2(ds, dt)
    B, H, D = x0.shape
    for solve in (1, 0):
        x, delta = x0.clone(), torch.zeros_like(x0)
        state = torch.zeros((B, 4), device="cuda"); state[:, 0], state[:, 1] = 3.0e38, opt.lambda_init
        st = _lib.current_stream()
        for _ in range(3):
            lib.mpdx_gpmp_step(C.byref(opt.gp), C.byref(opt.opts), x.data_ptr(), delta.data_ptr(), state.data_ptr(), B, H, D, solve, st)
        torch.cuda.synchronize()
        early = delta[0].clone()   # (-DMPDX_GPMP_STAMPS build: the stamps of trajectory 0's third iteration - it may have converged by the end of the timed loop)
        t0 = time.perf_counter()
        for _ in range(50):
            lib.mpdx_gpmp_step(C.byref(opt.gp), C.byref(opt.opts), x.data_ptr(), delta.data_ptr(), state.data_ptr(), B, H, D, solve, st)
        torch.cuda.synchronize()
        print(f"{env_id}-{robot_id}: solve={solve}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch (B={B}, early iterations: every trajectory active)")
        if solve and float(early[0].abs().max()) > 0:   # -DMPDX_GPMP_STAMPS build: s_memtime ticks of workgroup 0's phases
            tk = early[0, :4].tolist()
            print(f"   stamps (shader clocks): linearise+judge {tk[0]:.0f}  assemble {tk[1]:.0f}  solve {tk[2]:.0f}  write-back {tk[3]:.0f}")
            bs = early[0:4].reshape(-1)[8:48].tolist() if early.numel() >= 48 and early.shape[1] >= 12 else []
            if bs and any(bs):   # block cyclic reduction: after P1 / P2 / P3 of every level (slots 1 + 3 lev ..), 30 = forward done, 31 + lev = substitution level done
                fw = [(round(bs[1 + 3 * l]), round(bs[2 + 3 * l]), round(bs[3 + 3 * l])) for l in range(6)]
                print("   bcr stamps (ticks from solve start) per level (P1, P2, P3):", fw, " forward done", round(bs[30]), " substitution levels 5..0 done", [round(bs[31 + l]) for l in range(5, -1, -1)])
    for sig in (1e-3, 5e-4, 2e-4):
        for iters in (500, 1000):
            o = GPMP2(ds, dt, sigma_obs=sig)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            xo = o.optimize(x0, opt_iters=iters)
            torch.cuda.synchronize()
            print(f"   sigma_obs={sig:g} iters={iters}: fraction free {ds.task.compute_fraction_free_trajs(xo):.2f}  init free {ds.task.compute_fraction_free_trajs(x0):.2f}  {(time.perf_counter()-t0)*1e3:.0f} ms")
