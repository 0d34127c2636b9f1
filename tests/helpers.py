"""Shared test helpers: synthetic weights, toy costs (same formulas as tests/golden/make_golden.py)."""
import numpy as np
import torch

from mpd_public_amd import synthetic as syn
from oracle.unet import unet_param_shapes

DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}


def synth_sd(D, opt):
    return syn.synth_state_dict(unet_param_shapes(D, 32, DIM_MULTS[opt]))


# (horizon, state_dim, dim_mults option) of tests/golden/shapes.npz (make_golden.py::SHAPE_CASES) and the probe vector stored per case
SHAPE_CASES = ((32, 4, 1), (128, 4, 0), (48, 4, 1), (24, 6, 0), (40, 2, 1), (96, 14, 1), (64, 24, 0))


def grad_probe(grads, names):
    """16 strided samples of every gradient tensor, parameter order (make_golden.py::grad_probe)"""
    parts = []
    for k in names:
        v = grads[k].detach().reshape(-1).to(torch.float32).cpu()
        parts.append(v[::max(1, v.numel() // 16)][:16])
    return torch.cat(parts)


def shape_case_batch(H, D, opt, B=5):
    tag = f"H{H}_D{D}_opt{opt}"
    x0, noise = t(f"shp_x0_{tag}", (B, H, D), "uniform", 0.8), t(f"shp_noise_{tag}", (B, H, D))
    hc = {0: t(f"shp_hc0_{tag}", (B, D), "uniform", 0.7), H - 1: t(f"shp_hc1_{tag}", (B, D), "uniform", 0.7)}
    return tag, x0, noise, hc, torch.tensor([3, 24, 0, 12, 7])


def toy_cost(x, x_interpolated=None, return_invidual_costs_and_weights=False, **kw):
    q = x.shape[-1] // 2
    c1 = (x_interpolated[..., :q] - 0.3).pow(2).sum((-1, -2)) * 3.0
    c2 = (x[:, 1:, :] - x[:, :-1, :]).pow(2).sum((-1, -2)) * 0.5 + (x[..., q:]).abs().sum((-1, -2)) * 0.01
    return [c1, c2], [1e-2, 3e-3]


def t(name, shape, kind="normal", scale=1.0):
    return torch.from_numpy(syn.synth_tensor(name, shape, kind, scale))


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


# ---------------------------------------------------------------------------------------------- guidance helpers
def oracle_guide(dataset, w_coll=1e-2, w_smooth=1e-7, clip_grad=True, interpolate=True, n_interp=128, dtype=torch.float32,
                 clip_grad_rule="norm", max_grad_value=0.1, gp_half_factor=False):
    """The oracle's guide (autograd over oracle/costs.py) for the same task as a product `TrajectoryDataset`."""
    from oracle import costs as oc
    from oracle.guide import GuideManager
    from oracle.normalizer import LimitsNormalizer
    from mpd_public_amd import _lib
    env, rob = dataset.env, dataset.robot
    dim = env.dim
    robot = oc.RobotPanda() if rob.name == "RobotPanda" else oc.RobotPointMass(rob.q_dim, rob.link_margin)
    robot.radii = robot.radii.to(dtype)
    cl, wl = [], []
    for f in dataset.task.get_collision_fields():
        if f.kind == _lib.FIELD_OBJECTS:
            o = f.objects
            fld = oc.ObjectField(torch.tensor(o.sphere_centers[:, :dim], dtype=dtype), torch.tensor(o.sphere_radii, dtype=dtype),
                                 torch.tensor(o.box_centers[:, :dim], dtype=dtype), torch.tensor(o.box_half[:, :dim], dtype=dtype))
        elif f.kind == _lib.FIELD_WORKSPACE:
            fld = oc.WorkspaceField(torch.tensor(f.ws_min, dtype=dtype), torch.tensor(f.ws_max, dtype=dtype))
        else:
            fld = oc.SelfField(torch.tensor(oc.PANDA_SELF_PAIRS, dtype=torch.long))
        cl.append(oc.CostCollision(robot, 64, field=fld, sigma_coll=1.0, cutoff_margin=dataset.task.obstacle_cutoff_margin))
        wl.append(w_coll)
    dt = 5.0 / dataset.n_support_points
    cl.append(oc.CostGPTrajectory(robot, 64, dt, sigma_gp=1.0, half_factor=gp_half_factor))
    wl.append(w_smooth)
    comp = oc.CostComposite(robot, 64, cl, weights_cost_l=wl)
    if getattr(dataset.normalizer, "kind", "limits") == "identity":
        from oracle.normalizer import Identity
        nrm = Identity()
    elif getattr(dataset.normalizer, "kind", "limits") == "gaussian":
        from oracle.normalizer import GaussianNormalizer
        nrm = GaussianNormalizer(dataset.normalizer.means.cpu(), dataset.normalizer.stds.cpu())
        nrm.means, nrm.stds = nrm.means.to(dtype), nrm.stds.to(dtype)
    else:
        nrm = LimitsNormalizer(dataset.normalizer.mins.cpu(), dataset.normalizer.maxs.cpu())
        nrm.mins, nrm.maxs = nrm.mins.to(dtype), nrm.maxs.to(dtype)
    return GuideManager(nrm, comp, clip_grad=clip_grad, interpolate=interpolate, n_interp=n_interp, clip_grad_rule=clip_grad_rule,
                        max_grad_value=max_grad_value), comp


def product_guide(dataset, w_coll=1e-2, w_smooth=1e-7, clip_grad=True, interpolate=True, clip_grad_rule="norm", max_grad_value=0.1,
                  gp_half_factor=False):
    """The product guide built exactly as scripts/inference/inference.py:188-236 builds it."""
    import mpd_public_amd as m
    H = dataset.n_support_points
    dt = 5.0 / H
    costs = [m.CostCollision(dataset.robot, H, field=f, sigma_coll=1.0) for f in dataset.task.get_collision_fields()]
    weights = [w_coll] * len(costs)
    costs.append(m.CostGPTrajectory(dataset.robot, H, dt, sigma_gp=1.0, half_factor=gp_half_factor))
    weights.append(w_smooth)
    comp = m.CostComposite(dataset.robot, H, costs, weights_cost_l=weights)
    from math import ceil
    return m.GuideManagerTrajectoriesWithVelocity(dataset, comp, clip_grad=clip_grad, clip_grad_rule=clip_grad_rule, max_grad_value=max_grad_value,
                                                  interpolate_trajectories_for_collision=interpolate,
                                                  num_interpolated_points=ceil(H * 1.5))  # misspelt kwarg, as inference.py:234


def obstacle_hugging_trajs(dataset, B, seed="traj", scale=1.0):
    """Normalised [B,64,D] trajectories: straight lines between random configurations + noise, so that many waypoints
    sit inside obstacle margins / outside the workspace (every hinge branch is exercised)."""
    D = dataset.state_dim
    qd = D // 2
    from oracle.normalizer import LimitsNormalizer as _N
    cpu_norm = _N(dataset.normalizer.mins.cpu(), dataset.normalizer.maxs.cpu())
    a = t(f"{seed}/a", (B, 1, qd), "uniform", 0.95)
    b = t(f"{seed}/b", (B, 1, qd), "uniform", 0.95)
    if dataset.robot.name == "RobotPointMass":  # trajectory 0: from outside the workspace corner into an extra object
        a[0, 0, :] = -1.1
        c = torch.tensor(dataset.env.obj_extra.sphere_centers[0, :qd])
        b[0, 0, :] = cpu_norm.normalize(torch.cat([c, c]))[:qd]
    if dataset.robot.name == "RobotPanda":  # trajectory 0: from a self-colliding pose to a pose that leaves the workspace
        qa, qb = panda_probe_configs(dataset)
        nq = cpu_norm.normalize(torch.stack([torch.cat([qa, qa]), torch.cat([qb, qb])]))[:, :qd]
        a[0, 0, :], b[0, 0, :] = nq[0], nq[1]
    s = torch.linspace(0, 1, 64).reshape(1, 64, 1)
    pos = a + (b - a) * s + 0.05 * t(f"{seed}/n", (B, 64, qd))
    vel = 0.3 * t(f"{seed}/v", (B, 64, qd))
    return (scale * torch.cat([pos, vel], -1)).contiguous()


_PROBE = {}


def panda_probe_configs(dataset):
    """(q with an active self-collision hinge, q with an active workspace-boundary hinge), found by a deterministic
    scan of hash-uniform joint samples through the oracle's FK."""
    if "panda" in _PROBE:
        return _PROBE["panda"]
    from oracle import costs as oc
    rob = oc.RobotPanda()
    lo, hi = dataset.normalizer.mins[:7].cpu(), dataset.normalizer.maxs[:7].cpu()
    q = lo + (hi - lo) * (t("panda_probe", (4000, 7), "uniform") * 0.5 + 0.5)
    P = rob.link_points(q)
    pairs = torch.tensor(oc.PANDA_SELF_PAIRS)
    d = torch.linalg.norm(P[:, pairs[:, 0]] - P[:, pairs[:, 1]], dim=-1)
    selfc = torch.relu(rob.radii[pairs[:, 0]] + rob.radii[pairs[:, 1]] - d).sum(-1)
    m = (rob.radii + dataset.task.obstacle_cutoff_margin).unsqueeze(-1)
    wsc = (torch.relu(m - (P - torch.tensor(dataset.task.ws_min))) + torch.relu(m - (torch.tensor(dataset.task.ws_max) - P))).sum((-1, -2))
    assert selfc.max() > 0 and wsc.max() > 0
    _PROBE["panda"] = (q[selfc.argmax()], q[wsc.argmax()])
    return _PROBE["panda"]


def oracle_plan_metrics(dataset, xu, n_check=256):
    """The oracle's restatement of the post-loop metrics (inference.py:288-297,311-316) on UNNORMALISED trajectories xu [B,H,D]
    (CPU tensor): (#colliding interpolated waypoints [B], path length [B], smoothness [B]).  A waypoint collides iff some hinge
    with margin = link radius (no cutoff margin) is active on the n_check-point interpolation."""
    from oracle import costs as oc
    from oracle.guide import interpolate_points_v1
    _, comp = oracle_guide(dataset, dtype=xu.dtype)
    qd = dataset.state_dim // 2
    q, v = xu[..., :qd], xu[..., qd:]
    plen = torch.linalg.norm(q[:, 1:] - q[:, :-1], dim=-1).sum(-1)
    smooth = torch.linalg.norm(v[:, 1:] - v[:, :-1], dim=-1).sum(-1)
    xi = interpolate_points_v1(xu, n_check)
    hit = torch.zeros(xi.shape[:2], dtype=torch.bool)
    for term in comp.cost_l:
        if isinstance(term, oc.CostCollision):
            term.cutoff = 0.0
            per_point = torch.stack([term(xi[:, i:i + 1]) for i in range(xi.shape[1])], 1)
            hit |= per_point > 0
    return hit.sum(1), plen, smooth


import contextlib
import os


@contextlib.contextmanager
def kernel_path(fused: bool):
    """Run the enclosed product calls on the fused level programs (default) or on the per-layer conv kernels everywhere
    (MPDX_FUSED=0; libmpdx reads the variable on every pass)."""
    old = os.environ.get("MPDX_FUSED")
    os.environ["MPDX_FUSED"] = "1" if fused else "0"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("MPDX_FUSED", None)
        else:
            os.environ["MPDX_FUSED"] = old


# ---------------------------------------------------------------------------------------------- decidable plan figures (guided plans)
def oracle_hinge_slack(dataset, xu, n_check=256, dtype=torch.float64):
    """Per interpolated waypoint of the UNNORMALISED trajectories xu [B,H,D] (CPU): slack[b,i] = max over EVERY collision hinge of the
    waypoint (link sphere x field: objects / workspace faces / self-collision pairs) of (margin - signed distance), margin = link
    radius (no cutoff margin) - the quantity whose sign IS the collision flag of inference.py:288-297 (a waypoint collides iff
    slack > 0), evaluated in `dtype`.  |slack| < eps marks a waypoint whose flag no fp32 evaluation can decide."""
    from oracle import costs as oc
    from oracle.guide import interpolate_points_v1
    _, comp = oracle_guide(dataset, dtype=dtype)
    xi = interpolate_points_v1(xu.to(dtype), n_check)
    slack = torch.full(xi.shape[:2], -float("inf"), dtype=dtype)
    for term in comp.cost_l:
        if not isinstance(term, oc.CostCollision):
            continue
        rob, f = term.robot, term.field
        pts = rob.link_points(xi[..., : rob.q_dim])          # [B, N, K, dim]
        radii = rob.radii.to(dtype)
        if f.kind == "objects":
            s = radii - f.sdf(pts)
        elif f.kind == "workspace":
            m = radii.unsqueeze(-1)
            s = torch.cat([m - (pts - f.ws_min.to(dtype)), m - (f.ws_max.to(dtype) - pts)], dim=-1).flatten(-2)
        else:
            a, b = pts[..., f.pairs[:, 0], :], pts[..., f.pairs[:, 1], :]
            s = radii[f.pairs[:, 0]] + radii[f.pairs[:, 1]] - torch.linalg.norm(a - b, dim=-1)
        slack = torch.maximum(slack, s.amax(-1))
    return slack


def guided_parity_record(dm, sd, guide, gk, hc, T, n0, nb, n_check=256, eps=1e-5, seed=31, threads=16, weights=(1e-2, 1e-7)):
    """The DECIDABLE form of north_star's "collision-free rate and smoothness identical to 3 s.f." for guided plans (checker leg; used by
    tests/test_gpu_guided_class.py and by bench.py's guided.oracle_check).

    A guided chain is discontinuous: a waypoint within fp32 rounding of a hinge margin takes a clipped increment (w = 1e-2) in one
    evaluation and not in another, and 150 guide iterations + the U-Net spread the flip - so two CORRECT fp32 evaluations of one plan end
    ~1e-2 apart on every trajectory (measured: the fp32 CPU oracle against its own fp64 run), and the count of colliding waypoints of a
    small slice differs in the third figure between any two of them.  What can be decided, and is:
      (1) same plan - the HIP metrics kernel's per-waypoint collision flags on the HIP plan against an fp64 evaluation of the SAME
          trajectories: a waypoint is `ambiguous` iff its fp64 hinge slack max_h(margin_h - sdf_h) lies within +-eps of zero; every
          other waypoint must carry the same flag, so the figures over the unambiguous waypoints / trajectories agree EXACTLY
          (path length and smoothness: to fp32 rounding);
      (2) chain class - flags of the HIP plan, of the fp32 CPU oracle's plan and of the fp64 oracle's plan (same injected noise):
          flips(HIP vs fp64) <= 2 * flips(fp32 oracle vs fp64) + 2, i.e. the HIP chain is as close to the rounding-free chain as the
          oracle's own fp32 arithmetic is (the guided analogue of test_chain_error_is_fp32_rounding_class).
    Returns a JSON-able dict; `equal_to_3sf` is (1) per figure, `chain_class.within_fp32_class` is (2)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import diffusion as odiff
    from oracle import metrics as omet
    ds = guide.dataset
    D = ds.state_dim
    qd = D // 2
    noise = torch.randn((T + n0 + 1, nb, 64, D), generator=torch.Generator().manual_seed(seed))
    kw = dict(n_guide_steps=gk["n_guide_steps"], t_start_guide=gk["t_start_guide"], n_diffusion_steps_without_noise=n0)
    dev = next(iter(hc.values())).device   # (the product path is CUDA-only; the CPU test of THIS function injects stand-ins)
    chain = dm.run_inference(None, hc, n_samples=nb, horizon=64, return_chain=True, guide=guide, noise_std_extra_schedule_fn=lambda t: 0.5,
                             noise=noise.to(dev), **kw)
    xu_hip_dev = ds.unnormalize_trajectories(chain[-1])
    m, mask_hip = ds.task.trajectory_metrics(xu_hip_dev, n_check=n_check, return_mask=True)
    m, mask_hip, x_hip, xu_hip = m.cpu(), mask_hip.cpu(), chain[-1].cpu(), xu_hip_dev.cpu()
    hcc = {k: v.cpu() for k, v in hc.items()}
    old_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, old_threads)))   # (more threads are slower on this ATen loop: bench.py's cpu_baseline probe)

    def oracle_chain(dtype):
        og, _ = oracle_guide(ds, *weights, dtype=dtype)
        t0 = time.perf_counter()
        r = odiff.run_inference({k: v.to(dtype) for k, v in sd.items()}, {k: v.to(dtype) for k, v in hcc.items()}, noise.to(dtype), T,
                                noise_std=0.5, guide=og, dtype=dtype, **kw)
        return r[-1], og.normalizer.unnormalize(r[-1]), time.perf_counter() - t0
    try:
        with ThreadPoolExecutor(2) as ex:
            f32, f64 = ex.submit(oracle_chain, torch.float32), ex.submit(oracle_chain, torch.float64)
            (x32, xu32, s32), (x64, xu64, s64) = f32.result(), f64.result()
    finally:
        torch.set_num_threads(old_threads)

    def figures(hit, xu):      # per-plan figures from per-waypoint flags [nb, n_check] + fp64 path metrics
        z = xu.double().numpy()
        return {"collision_free_rate": float((hit.sum(1) == 0).float().mean()), "collision_intensity": float(hit.float().mean(1).mean()),
                "path_length": float(omet.compute_path_length(z, qd).mean()), "smoothness": float(omet.compute_smoothness(z, qd).mean())}

    def same3(a, b):           # identical to 3 significant figures (or both zero)
        return a == b or abs(a - b) <= 5e-3 * max(abs(a), abs(b))

    # ---- (1) same plan: HIP kernel vs fp64 evaluation of the HIP trajectories
    slack = oracle_hinge_slack(ds, xu_hip, n_check)
    hit64, amb = slack > 0, slack.abs() < eps
    dec = ~amb
    disagree = int(((mask_hip != hit64) & dec).sum())
    traj_dec = ~((amb & ~((mask_hip & dec).any(1, keepdim=True))).any(1))   # a trajectory whose only possible hits are ambiguous has no decidable "free" status

    def unamb(hit):
        frac = (hit & dec).sum(1).double() / dec.sum(1).clamp(min=1).double()
        free = ((hit & dec).sum(1) == 0)[traj_dec]
        return {"collision_free_rate": float(free.float().mean()) if free.numel() else float("nan"), "collision_intensity": float(frac.mean())}
    hip_u, f64_u = unamb(mask_hip), unamb(hit64)
    hip_u.update(path_length=float(m[:, 1].mean()), smoothness=float(m[:, 2].mean()))
    z = xu_hip.double().numpy()
    f64_u.update(path_length=float(omet.compute_path_length(z, qd).mean()), smoothness=float(omet.compute_smoothness(z, qd).mean()))
    equal = {"collision_free_rate": bool(disagree == 0 and hip_u["collision_free_rate"] == f64_u["collision_free_rate"]),
             "collision_intensity": bool(disagree == 0 and hip_u["collision_intensity"] == f64_u["collision_intensity"]),
             "path_length": bool(abs(hip_u["path_length"] - f64_u["path_length"]) <= 2e-5 * abs(f64_u["path_length"])),
             "smoothness": bool(abs(hip_u["smoothness"] - f64_u["smoothness"]) <= 2e-5 * abs(f64_u["smoothness"]))}
    # ---- (2) chain class: final flags of the three chains
    hit32c, hit64c = oracle_hinge_slack(ds, xu32, n_check) > 0, oracle_hinge_slack(ds, xu64, n_check) > 0
    flips_hip, flips_32 = int((mask_hip != hit64c).sum()), int((hit32c != hit64c).sum())
    div = lambda a: int(((a.double() - x64).abs().amax((1, 2)) > 1e-3).sum())
    fig = {"hip": {"collision_free_rate": float((m[:, 0] == 0).float().mean()), "collision_intensity": float((m[:, 0] / m[:, 3]).mean()),
                   "path_length": float(m[:, 1].mean()), "smoothness": float(m[:, 2].mean())},
           "oracle_fp32": figures(hit32c, xu32), "oracle_fp64": figures(hit64c, xu64)}
    r6 = lambda d: {k: float(f"{v:.6g}") for k, v in d.items()}
    return {
        "trajectories": nb, "waypoints_checked": int(nb * n_check), "eps": eps,
        "definition": "equal_to_3sf = the HIP metrics kernel against an fp64 evaluation of the SAME (HIP) plan over the waypoints whose fp64 hinge slack is "
                      "not within +-eps of zero (ambiguous_waypoints excluded; counts must agree exactly, path length / smoothness to 2e-5); chain_class = "
                      "hinge flips of the final plan against the fp64 oracle chain, HIP vs the fp32 CPU oracle (tests/helpers.py::guided_parity_record)",
        "ambiguous_waypoints": int(amb.sum()), "undecidable_trajectories": int((~traj_dec).sum()), "flag_disagreements_outside_ambiguous": disagree,
        "same_plan": {"hip_kernel": r6(hip_u), "fp64_evaluation": r6(f64_u)},
        "equal_to_3sf": equal,
        "chain_class": {"flips_vs_fp64_chain": {"hip": flips_hip, "oracle_fp32": flips_32}, "bound": "hip <= 2 * oracle_fp32 + 2",
                        "within_fp32_class": bool(flips_hip <= 2 * flips_32 + 2),
                        "trajectories_diverged_gt_1e-3_from_fp64_chain": {"hip": div(x_hip), "oracle_fp32": div(x32)},
                        "max_abs_diff_final_trajectories": {"hip_vs_fp64": float((x_hip.double() - x64).abs().max()),
                                                            "oracle_fp32_vs_fp64": float((x32.double() - x64).abs().max()),
                                                            "hip_vs_oracle_fp32": float((x_hip - x32).abs().max())}},
        "plan_figures": {k: r6(v) for k, v in fig.items()},
        "cross_chain_equal_to_3sf": {"hip_vs_oracle_fp64": {k: bool(same3(fig["hip"][k], fig["oracle_fp64"][k])) for k in fig["hip"]},
                                     "oracle_fp32_vs_oracle_fp64": {k: bool(same3(fig["oracle_fp32"][k], fig["oracle_fp64"][k])) for k in fig["hip"]},
                                     "note": "different chains: the third figure of a count over nb x n_check waypoints moves between ANY two fp32 evaluations "
                                             "(second row: the CPU oracle against itself in fp64) - decided by chain_class instead"},
        "oracle_cpu_plan_s": {"fp32": round(s32, 2), "fp64": round(s64, 2)}, "weights": "synthetic (random-init): the figures are those of un-trained plans"}
