"""experiment() - drop-in for scripts/inference/inference.py:34-352 (the planning entry).

Same keyword arguments and defaults, same result dictionary / pickle keys (including the reference's swapped
'cost_path_length_trajs_final_free' <-> 'cost_smoothness_trajs_final_free', inference.py:345-346).  Differences, all
forced by what the reference tree does not contain:
  * trained weights / args.yaml / dataset are Google-Drive downloads (README.md:69-72).  If `model_dir` holds
    `args.yaml` + `checkpoints/{ema_,}model_current_state_dict.pth` they are loaded (state-dict keys are identical);
    otherwise formula-defined synthetic weights are used and `model_id` picks env/robot by name.
  * `experiment_launcher` decorators are replaced by plain kwargs; rendering (inference.py:356-434) is out of scope.
"""
from __future__ import annotations

import os
import pickle
import time
from math import ceil

import torch

from . import synthetic as syn
from .datasets import TrajectoryDataset
from .diffusion_model import GaussianDiffusionModel
from .guides import GuideManagerTrajectoriesWithVelocity
from .planning import CostCollision, CostComposite, CostGPTrajectory, compute_path_length, compute_smoothness, compute_variance_waypoints
from .sample_functions import ddpm_sample_fn, guide_gradient_steps
from .temporal_unet import TemporalUnet, UNET_DIM_MULTS


def _synthetic_args():
    # launch_train_01.py:51-84 is the only record of trained hyper-parameters
    return dict(variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True, unet_input_dim=32,
                unet_dim_mults_option=1, use_ema=True, include_velocity=True)


def experiment(model_id: str = "EnvSpheres3D-RobotPanda", planner_alg: str = "mpd", use_guide_on_extra_objects_only: bool = False,
               n_samples: int = 50, start_guide_steps_fraction: float = 0.25, n_guide_steps: int = 5,
               n_diffusion_steps_without_noise: int = 5, weight_grad_cost_collision: float = 1e-2,
               weight_grad_cost_smoothness: float = 1e-7, factor_num_interpolated_points_for_collision: float = 1.5,
               trajectory_duration: float = 5.0, device: str = "cuda", debug: bool = True, render: bool = False, seed: int = 30,
               results_dir: str = "logs", model_dir: str = None, model_args: dict = None, **kwargs):
    torch.manual_seed(seed)
    if not torch.cuda.is_available():
        raise RuntimeError("mpd_public_amd.inference needs an AMD GPU (no CPU fallback)")
    tensor_args = {"device": torch.device(device), "dtype": torch.float32}
    if planner_alg not in ("mpd", "diffusion_prior_then_guide", "diffusion_prior"):
        raise NotImplementedError(planner_alg)
    run_prior_only = planner_alg == "diffusion_prior"
    run_prior_then_guidance = planner_alg == "diffusion_prior_then_guide"

    args = _synthetic_args()
    ckpt = None
    if model_dir is not None and os.path.exists(os.path.join(model_dir, "args.yaml")):
        import yaml
        with open(os.path.join(model_dir, "args.yaml")) as f:
            args.update(yaml.safe_load(f))
        ckpt = os.path.join(model_dir, "checkpoints", "ema_model_current_state_dict.pth" if args.get("use_ema") else "model_current_state_dict.pth")
    if model_args:
        args.update(model_args)
    env_id, robot_id = model_id.split("-")

    dataset = TrajectoryDataset(env_id=env_id, robot_id=robot_id, use_extra_objects=True, obstacle_cutoff_margin=0.05,
                                include_velocity=args["include_velocity"], tensor_args=tensor_args)
    n_support_points, robot, task = dataset.n_support_points, dataset.robot, dataset.task
    dt = trajectory_duration / n_support_points
    robot.dt = dt

    unet = TemporalUnet(state_dim=dataset.state_dim, n_support_points=n_support_points, unet_input_dim=args["unet_input_dim"],
                        dim_mults=UNET_DIM_MULTS[args["unet_dim_mults_option"]])
    model = GaussianDiffusionModel(model=unet, variance_schedule=args["variance_schedule"], n_diffusion_steps=args["n_diffusion_steps"],
                                   predict_epsilon=args["predict_epsilon"])
    if ckpt is not None and os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
        # The reference derives the normaliser limits from the training dataset (normalization.py:92-93) and the obstacles /
        # link spheres from torch_robotics; neither travels with a checkpoint.  `model_dir/limits.yaml`
        # ({mins: [D], maxs: [D]}) supplies the limits; without it trained weights would be run against SYNTHETIC
        # normalisation - refuse unless the caller opts in.
        lim = os.path.join(model_dir, "limits.yaml")
        if os.path.exists(lim):
            import yaml
            with open(lim) as f:
                lm = yaml.safe_load(f)
            from .datasets import LimitsNormalizer, GaussianNormalizer, Identity
            if len(lm["mins"]) != dataset.state_dim or len(lm["maxs"]) != dataset.state_dim:
                raise ValueError(f"{lim}: expected {dataset.state_dim} mins/maxs")
            kind = lm.get("normalizer", "LimitsNormalizer")   # (written by train.py of this package; the limits are final: a Safe / Fixed one is a LimitsNormalizer here)
            if kind == "GaussianNormalizer":
                dataset.normalizer = GaussianNormalizer(lm["means"], lm["stds"], lm["mins"], lm["maxs"]).to(tensor_args["device"])
            elif kind == "Identity":
                dataset.normalizer = Identity(lm["mins"], lm["maxs"]).to(tensor_args["device"])
            else:
                dataset.normalizer = LimitsNormalizer(lm["mins"], lm["maxs"]).to(tensor_args["device"])
        elif not kwargs.get("allow_synthetic_limits", False):
            raise RuntimeError(f"{model_dir} holds trained weights but no limits.yaml: the normaliser limits (and the environment "
                               "geometry) of this package are synthetic stand-ins (DESIGN.md section 5) and would not match the "
                               "training data.  Provide model_dir/limits.yaml {mins, maxs} or pass allow_synthetic_limits=True.")
        else:
            import warnings
            warnings.warn("trained checkpoint combined with SYNTHETIC normaliser limits / environment geometry: plans and metrics "
                          "are not comparable with the reference's", RuntimeWarning)
    else:
        unet.load_state_dict(syn.synth_state_dict({k: tuple(v.shape) for k, v in unet.state_dict().items()}))
    model = model.to(tensor_args["device"]).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    model.manual_seed(seed)
    model.warmup(horizon=n_support_points, device=device)

    gen = torch.Generator(device=tensor_args["device"]).manual_seed(seed)
    start_state_pos = goal_state_pos = None
    for _ in range(100):
        q_free = task.random_coll_free_q(n_samples=2, device=tensor_args["device"], generator=gen)
        start_state_pos, goal_state_pos = q_free[0], q_free[1]
        if torch.linalg.norm(start_state_pos - goal_state_pos) > dataset.threshold_start_goal_pos:
            break
    if start_state_pos is None or goal_state_pos is None:
        raise ValueError("No collision free configuration was found")

    hard_conds = dataset.get_hard_conditions(torch.vstack((start_state_pos, goal_state_pos)), normalize=True)
    collision_fields = task.get_collision_fields_extra_objects() if use_guide_on_extra_objects_only else task.get_collision_fields()
    cost_l = [CostCollision(robot, n_support_points, field=f, sigma_coll=1.0, tensor_args=tensor_args) for f in collision_fields]
    weights = [weight_grad_cost_collision] * len(cost_l)
    cost_l.append(CostGPTrajectory(robot, n_support_points, dt, sigma_gp=1.0, tensor_args=tensor_args))
    weights.append(weight_grad_cost_smoothness)
    cost_composite = CostComposite(robot, n_support_points, cost_l, weights_cost_l=weights, tensor_args=tensor_args)
    guide = GuideManagerTrajectoriesWithVelocity(dataset, cost_composite, clip_grad=True, interpolate_trajectories_for_collision=True,
                                                 num_interpolated_points=ceil(n_support_points * factor_num_interpolated_points_for_collision),
                                                 tensor_args=tensor_args).to(tensor_args["device"])
    t_start_guide = ceil(start_guide_steps_fraction * model.n_diffusion_steps)
    sample_fn_kwargs = dict(guide=None if (run_prior_then_guidance or run_prior_only) else guide, n_guide_steps=n_guide_steps,
                            t_start_guide=t_start_guide, noise_std_extra_schedule_fn=lambda x: 0.5)

    t_cold = None
    if kwargs.get("warm_plan", False):
        # extension, OFF by default (so that `t_total` below covers what the reference's does: the first plan of the process, inference.py:248-259):
        # warm_plan=True runs one throwaway plan of the SAME shape in front of the timed one.  model.warmup() above mirrors the reference's (one U-Net
        # pass at batch 2); the first plan of a process additionally loads every kernel's code object, sizes the workspaces for n_samples, raises LDS
        # limits and uploads the guide's tables - 5 ... 60 ms (measured: 12.2 / 67 ms for the first plan against 7.3 / 6.6 ms for the next ones).
        # Both times are recorded (`t_total_cold` beside `t_total`); the noise stream is re-seeded: the timed plan is bit for bit the one an
        # un-warmed call would have produced.
        torch.cuda.synchronize()
        tc = time.perf_counter()
        model.run_inference(None, hard_conds, n_samples=n_samples, horizon=n_support_points, return_chain=True, sample_fn=ddpm_sample_fn,
                            **sample_fn_kwargs, n_diffusion_steps_without_noise=n_diffusion_steps_without_noise)
        torch.cuda.synchronize()
        t_cold = time.perf_counter() - tc
        model.manual_seed(seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trajs_normalized_iters = model.run_inference(None, hard_conds, n_samples=n_samples, horizon=n_support_points, return_chain=True,
                                                 sample_fn=ddpm_sample_fn, **sample_fn_kwargs,
                                                 n_diffusion_steps_without_noise=n_diffusion_steps_without_noise)
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    if debug:
        print(f"t_model_sampling: {t_total:.3f} sec")

    if run_prior_then_guidance:
        n_post = (t_start_guide + n_diffusion_steps_without_noise) * n_guide_steps
        hc_b = {k: v.reshape(1, -1).expand(n_samples, -1).contiguous() for k, v in hard_conds.items()}
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trajs, post = trajs_normalized_iters[-1].clone(), []
        for _ in range(n_post):
            trajs = guide_gradient_steps(trajs, hard_conds=hc_b, guide=guide, n_guide_steps=1, unnormalize_data=False)
            post.append(trajs)
        trajs_normalized_iters = torch.cat((trajs_normalized_iters, torch.stack(post, dim=0)))
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t1

    trajs_iters = dataset.unnormalize_trajectories(trajs_normalized_iters)
    trajs_final = trajs_iters[-1]
    trajs_final_coll, trajs_final_coll_idxs, trajs_final_free, trajs_final_free_idxs, _ = task.get_trajs_collision_and_free(trajs_final, return_indices=True)
    success_free_trajs = task.compute_success_free_trajs(trajs_final)
    fraction_free_trajs = task.compute_fraction_free_trajs(trajs_final)
    collision_intensity_trajs = task.compute_collision_intensity_trajs(trajs_final)
    if debug:
        print(f"success: {success_free_trajs}\npercentage free trajs: {fraction_free_trajs*100:.2f}\n"
              f"percentage collision intensity: {collision_intensity_trajs*100:.2f}")

    traj_final_free_best = idx_best_traj = cost_best_free_traj = cost_smoothness = cost_path_length = cost_all = None
    variance_waypoint_trajs_final_free = None
    if trajs_final_free is not None:
        cost_smoothness = compute_smoothness(trajs_final_free, robot)
        cost_path_length = compute_path_length(trajs_final_free, robot)
        cost_all = cost_path_length + cost_smoothness
        idx_best_traj = torch.argmin(cost_all).item()
        traj_final_free_best = trajs_final_free[idx_best_traj]
        cost_best_free_traj = torch.min(cost_all).item()
        variance_waypoint_trajs_final_free = compute_variance_waypoints(trajs_final_free, robot)
        if debug:
            print(f"cost smoothness: {cost_smoothness.mean():.4f}  cost path length: {cost_path_length.mean():.4f}  cost best: {cost_best_free_traj:.3f}")

    results_data_dict = {
        "trajs_iters": trajs_iters, "trajs_final_coll": trajs_final_coll, "trajs_final_coll_idxs": trajs_final_coll_idxs,
        "trajs_final_free": trajs_final_free, "trajs_final_free_idxs": trajs_final_free_idxs,
        "success_free_trajs": success_free_trajs, "fraction_free_trajs": fraction_free_trajs,
        "collision_intensity_trajs": collision_intensity_trajs, "idx_best_traj": idx_best_traj,
        "traj_final_free_best": traj_final_free_best, "cost_best_free_traj": cost_best_free_traj,
        "cost_path_length_trajs_final_free": cost_smoothness,   # sic: swapped in the reference (inference.py:345-346)
        "cost_smoothness_trajs_final_free": cost_path_length,
        "cost_all_trajs_final_free": cost_all, "variance_waypoint_trajs_final_free": variance_waypoint_trajs_final_free,
        "t_total": t_total,
    }
    if t_cold is not None:   # warm_plan=True: `t_total` above timed a warmed plan; the first plan of the process (what the reference's t_total covers) took this
        results_data_dict["t_total_cold"] = t_cold
    if results_dir:
        out_dir = os.path.join(results_dir, model_id, "results_inference", str(seed))
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "results_data_dict.pickle"), "wb") as handle:
            pickle.dump(results_data_dict, handle, protocol=pickle.HIGHEST_PROTOCOL)
    if render:
        raise NotImplementedError("rendering (inference.py:356-434: matplotlib videos + IsaacGym) is out of scope")
    return results_data_dict


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_id", default="EnvSpheres3D-RobotPanda")
    ap.add_argument("--planner_alg", default="mpd")
    ap.add_argument("--n_samples", type=int, default=50)
    ap.add_argument("--seed", type=int, default=30)
    ap.add_argument("--results_dir", default="logs")
    ap.add_argument("--model_dir", default=None, help="a results directory of mpd_public_amd.train (args.yaml, limits.yaml, checkpoints/): plan with its trained weights")
    ap.add_argument("--n_diffusion_steps_without_noise", type=int, default=5)
    a = ap.parse_args()
    experiment(**vars(a))
