"""End-to-end on one MI355X: generate_trajectories (RRT-Connect + GPMP2 on the device) -> train (native training step) -> guided planning with the
TRAINED weights through the inference entry.  Prints one JSON record: stage times, training loss, and the plan figures inference.py:288-327 reports
(fraction of collision-free trajectories, collision intensity, smoothness, path length) for the three planner_alg modes.

The point of it: with formula-defined weights every plan collides (free rate 0.0 on both sides of every parity check); with weights trained here the
figures north_star names are non-trivial.  python tools/trained_demo.py [--env EnvDense2D] [--contexts 64] [--per 16] [--steps 3000]"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="EnvDense2D")
    ap.add_argument("--robot", default="RobotPointMass")
    ap.add_argument("--contexts", type=int, default=64)
    ap.add_argument("--per", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--T", type=int, default=25)
    ap.add_argument("--opt", type=int, default=1)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    from mpd_public_amd import train as train_script
    from mpd_public_amd.generate_trajectories import generate_collision_free_trajectories as gen
    from mpd_public_amd.inference import experiment as infer
    root = Path(a.dir or tempfile.mkdtemp(prefix="mpdx_demo_"))
    sub = f"{a.env}-{a.robot}"
    rec = {"model_id": sub, "contexts": a.contexts, "trajectories_per_context": a.per, "train_steps": a.steps, "batch": a.batch, "T": a.T}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_free = 0
    for ctx in range(a.contexts):
        d = root / "data_trajectories" / sub / str(ctx)
        d.mkdir(parents=True, exist_ok=True)
        _, nf = gen(a.env, a.robot, a.per, str(d), gpmp_opt_iters=300, seed=ctx)
        n_free += nf
    torch.cuda.synchronize()
    rec["generate_s"] = round(time.perf_counter() - t0, 2)
    rec["collision_free_training_trajectories"] = int(n_free)
    t0 = time.perf_counter()
    logs = root / "logs"
    model, ema, losses = train_script.experiment(dataset_subdir=sub, data_dir=str(root / "data_trajectories"), results_dir=str(logs), n_diffusion_steps=a.T,
                                                 unet_dim_mults_option=a.opt, batch_size=a.batch, lr=a.lr, num_train_steps=a.steps,
                                                 steps_til_summary=max(50, a.steps // 10), steps_til_ckpt=a.steps, seed=1, summary_class=None, debug=False)
    torch.cuda.synchronize()
    rec["train_s"] = round(time.perf_counter() - t0, 2)
    vals = [float(v["diffusion_loss"]) for _, v in losses if "diffusion_loss" in v]
    rec["loss_first_last"] = [round(vals[0], 5), round(vals[-1], 5)] if vals else None
    out = {}
    for alg in ("diffusion_prior", "mpd", "diffusion_prior_then_guide"):
        rows = []
        for seed in range(a.seeds):
            r = infer(model_id=sub, planner_alg=alg, model_dir=str(logs), n_samples=a.samples, debug=False, results_dir=None, seed=30 + seed)
            rows.append({"fraction_free": float(r["fraction_free_trajs"]), "collision_intensity": float(r["collision_intensity_trajs"]),
                         "success": int(r["success_free_trajs"]), "t_total_ms": round(1e3 * float(r["t_total"]), 2),
                         "cost_best_free_traj": None if r["cost_best_free_traj"] is None else round(float(r["cost_best_free_traj"]), 4)})
        out[alg] = rows
    rec["plans"] = out
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
