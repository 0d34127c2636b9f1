#!/usr/bin/env python3
"""bench.py - denoising-steps/s and plan wall-clock of the reverse-diffusion planning loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run,
one rank per GPU.  A "step" here is ONE FULL PLAN = the region the reference times (scripts/inference/inference.py:
248-258): initial noise, all T+5 reverse updates of the whole batch, chain materialisation.
  metric  = denoising-steps/s = (T+5) * plans / wall ; ms_per_step = plan wall-clock in ms.
Workload (BASELINE.json configs[1]): EnvDense2D-RobotPointMass shape - 100 trajectories x H=64 x D=4 (pos+vel),
T=100 diffusion steps (+5 without noise), U-Net dim_mults (1,2,4,8), unguided, fp32, synthetic formula-defined weights.
N>1: the single-context plan does not shard (B=100 does not fill a GPU): N independent replicas, no collective
in the data path (SURVEY.md 8e "replicas only"); value is the aggregate over ranks.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

CONFIGS = {
    # name: (env, robot, D, dim_mults, T, B, n_without_noise, guided, n_contexts)
    "cfg1": ("EnvSimple2D", "RobotPointMass", 4, (1, 2, 4, 8), 25, 8, 5, False, 1),
    "cfg2": ("EnvDense2D", "RobotPointMass", 4, (1, 2, 4, 8), 100, 100, 5, False, 1),          # BASELINE configs[1]: the metric's config
    "cfg3": ("EnvNarrowPassageDense2D", "RobotPointMass", 4, (1, 2, 4, 8), 100, 100, 5, True, 1),
    "cfg4": ("EnvSpheres3D", "RobotPanda", 14, (1, 2, 4, 8), 100, 100, 5, True, 1),
    "cfg5": ("EnvSpheres3D", "RobotPanda", 14, (1, 2, 4, 8), 100, 6400, 5, True, 128),        # per-GPU shard of 1024 ctx x 50
}
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_16x16x4_f32) = fp32 vector peak
HBM_PEAK_GBS = 8000.0


def build_model(D, mults, T, device):
    import mpd_public_amd as m
    from mpd_public_amd import synthetic as syn
    net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=mults)
    sd = syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})  # formula-defined weights
    net.load_state_dict(sd, strict=True)
    dm = m.GaussianDiffusionModel(model=net, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
    return dm.to(device).eval(), sd


def roofline_leg(dm, B, T, reps=20):
    """Per-launch durations of one U-Net pass measured with HIP event pairs on the launch stream (mpdx_unet_profile),
    averaged over `reps` passes.  Returns the roofline object for the dominant kernel class + a per-class table."""
    from mpd_public_amd import _lib
    lib = _lib.load()
    hdl, packed, tab, ws = dm.model.engine(T, B)
    x = torch.randn(B, 64, dm.state_dim, device="cuda")
    cap = 128
    ms = (C.c_float * cap)()
    fl = (C.c_double * cap)()
    names = (C.c_char_p * cap)()
    n = C.c_int()
    acc = None
    st = torch.cuda.current_stream().cuda_stream
    for r in range(reps + 2):
        _lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), T // 2, B,
                                         ws.data_ptr(), st, cap, ms, fl, names, C.byref(n)), "mpdx_unet_profile")
        if r < 2:
            continue
        cur = [ms[i] for i in range(n.value)]
        acc = cur if acc is None else [a + b for a, b in zip(acc, cur)]
    avg_ms = [a / reps for a in acc]
    buf = C.create_string_buffer(64)
    classes = defaultdict(lambda: [0.0, 0.0, 0])  # key -> [ms, flops, launches]
    for i in range(n.value):
        nm = names[i].decode()
        li = lib.mpdx_unet_unit_layer(hdl, B, i)
        if li >= 0:
            lib.mpdx_unet_layer_tile(hdl, li, B, buf, 64)
            kind = "conv_k5_gn_mish" if ".block.0." in nm else ("conv_k1" if "residual" in nm else ("down_k3s2" if "downs" in nm else "up_k4s2"))
            key = f"{kind}[{buf.value.decode()}] flops/launch={fl[i]:.3e}"
        else:
            key = nm
        c = classes[key]
        c[0] += avg_ms[i]; c[1] += fl[i]; c[2] += 1
    table = sorted(((k, v[0], v[1], v[2]) for k, v in classes.items()), key=lambda r: -r[1])
    dom = max((r for r in table if r[2] > 0 and not r[0].startswith("fused")), key=lambda r: r[1])
    per_launch_flops = dom[2] / dom[3]
    # The per-launch event pairs above carry the event records' own cost (14.6 us vs 11.8 us under rocprofv3 for the
    # dominant class), and back-to-back launches of one layer re-read warm weights (10.3 us).  For the roofline number the
    # dominant class is timed IN SITU: one HIP-event pair on the launch stream brackets the longest run of consecutive
    # launches of that class inside real U-Net passes (6 launches for the 256->256 blocks), 50 passes.  This is the
    # figure that agrees with the rocprofv3 --kernel-trace --stats average committed under profiles/.
    keys = []
    for i in range(n.value):
        li = lib.mpdx_unet_unit_layer(hdl, B, i)
        k = None
        if li >= 0:
            lib.mpdx_unet_layer_tile(hdl, li, B, buf, 64)
            nm = names[i].decode()
            kind = "conv_k5_gn_mish" if ".block.0." in nm else ("conv_k1" if "residual" in nm else ("down_k3s2" if "downs" in nm else "up_k4s2"))
            k = f"{kind}[{buf.value.decode()}] flops/launch={fl[i]:.3e}"
        keys.append(k)
    best = (0, 0, -1)
    i = 0
    while i < len(keys):
        if keys[i] == dom[0]:
            j = i
            while j + 1 < len(keys) and keys[j + 1] == dom[0]:
                j += 1
            if j - i + 1 > best[0]:
                best = (j - i + 1, i, j)
            i = j + 1
        else:
            i += 1
    out = C.c_float()
    if best[0] > 0:
        _lib.check(lib.mpdx_unet_time_units(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), T // 2, B,
                                            ws.data_ptr(), st, best[1], best[2], 50, C.byref(out)), "mpdx_unet_time_units")
        per_launch_ms = out.value / best[0]
    else:
        per_launch_ms = dom[1] / dom[3]
    achieved = per_launch_flops / (per_launch_ms * 1e-3) / 1e12
    # HBM-side traffic per launch of the dominant kernel: PMC counters cannot be collected from inside this process;
    # the value measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction) is
    # committed under profiles/ and quoted here when it is for this kernel and batch.
    traffic = None
    pmc = ROOT / "profiles" / "r01_pmc_dominant.json"
    if pmc.exists() and B == 100 and "32x32/1x8" in dom[0] and "conv_k5_gn_mish" in dom[0]:
        traffic = json.loads(pmc.read_text()).get("traffic_bytes_per_launch")
    roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
            "kernel": dom[0], "launches_per_unet_pass": dom[3], "avg_launch_us": round(per_launch_ms * 1e3, 2),
            "timed": f"in situ, one event pair around {best[0]} consecutive launches, 50 passes",
            "unet_pass_us_sum_of_launches": round(sum(avg_ms) * 1e3, 1),
            "unet_pass_tflops": round(sum(fl[i] for i in range(n.value)) / (sum(avg_ms) * 1e-3) / 1e12, 3)}
    return roof, table, sum(fl[i] for i in range(n.value))


def cpu_baseline_leg(sd, D, T, B, n0, min_seconds=10.0, max_plans=12):
    """The CPU oracle (oracle/, a port of the reference's algorithm validated against it) timed on this box's host
    cores on the SAME workload: whole plans until >= min_seconds of CPU work (bounded sample)."""
    from oracle import diffusion as odiff
    from mpd_public_amd import synthetic as syn
    # thread count: torch's default (all logical cores) oversubscribes these small per-layer ops badly on big hosts
    # (measured: 128 threads -> 5.9 steps/s, slower than 8 threads); probe a few settings on 2 steps and keep the best.
    import os as _os
    probe_noise = torch.randn((4, B, 64, D))
    hc0 = {0: torch.zeros(D), 63: torch.zeros(D)}
    best = (1e30, torch.get_num_threads())
    for nthr in sorted({8, 16, 32, 64, min(_os.cpu_count() or 8, 128)}):
        if nthr > (_os.cpu_count() or 8):
            continue
        torch.set_num_threads(nthr)
        odiff.run_inference(sd, hc0, probe_noise, 2, n_diffusion_steps_without_noise=1, noise_std=0.5)
        t0 = time.perf_counter()
        odiff.run_inference(sd, hc0, probe_noise, 2, n_diffusion_steps_without_noise=1, noise_std=0.5)
        dt_ = time.perf_counter() - t0
        if dt_ < best[0]:
            best = (dt_, nthr)
    torch.set_num_threads(best[1])
    cores = best[1]
    hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform")), 63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform"))}
    gen = torch.Generator().manual_seed(30)
    noise = torch.randn((T + n0 + 1, B, 64, D), generator=gen)
    odiff.run_inference(sd, hc, noise[:4], 2, n_diffusion_steps_without_noise=1, noise_std=0.5)  # warm-up (3 steps)
    plans, t0 = 0, time.perf_counter()
    while plans < max_plans and (time.perf_counter() - t0 < min_seconds or plans == 0):
        odiff.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5)
        plans += 1
    dt = time.perf_counter() - t0
    return {"value": round(plans * (T + n0) / dt, 2), "unit": "denoising-steps/s", "cores": cores, "kind": "port",
            "sample": f"{plans} full plan(s) of {T + n0} steps, B={B}, torch-CPU fp32 oracle, {dt:.1f} s",
            "plan_wall_s": round(dt / plans, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL over xGMI
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    env_id, robot, D, mults, T, B, n0, guided, n_ctx = CONFIGS[args.config]
    dm, sd = build_model(D, mults, T, f"cuda:{local_rank}")
    dm.manual_seed(30 + rank)
    from mpd_public_amd import synthetic as syn
    import mpd_public_amd as m
    from math import ceil
    extra = lambda t: 0.5  # noqa: E731  inference.py:243
    guide_kw = {}
    if guided:  # guide built exactly as inference.py:188-236 builds it (weights 1e-2 / 1e-7, 5 guide steps, last quarter)
        ds = m.TrajectoryDataset(env_id, robot, tensor_args={"device": torch.device("cuda", local_rank), "dtype": torch.float32})
        H_, dt_ = 64, 5.0 / 64
        cl = [m.CostCollision(ds.robot, H_, field=f, sigma_coll=1.0) for f in ds.task.get_collision_fields()]
        wl = [1e-2] * len(cl)
        cl.append(m.CostGPTrajectory(ds.robot, H_, dt_, sigma_gp=1.0)); wl.append(1e-7)
        guide = m.GuideManagerTrajectoriesWithVelocity(ds, m.CostComposite(ds.robot, H_, cl, weights_cost_l=wl), clip_grad=True,
                                                       interpolate_trajectories_for_collision=True).cuda()
        guide_kw = dict(guide=guide, n_guide_steps=5, t_start_guide=ceil(0.25 * T))
    if n_ctx == 1:
        hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform", 0.6)).cuda(),
              63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform", 0.6)).cuda()}

        def one_plan():
            return dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=True, n_diffusion_steps_without_noise=n0,
                                    noise_std_extra_schedule_fn=extra, **guide_kw)
    else:  # per-rank shard of independent contexts; per-trajectory hard conditions
        from mpd_public_amd.parallel import expand_contexts, gather_trajectories
        st = torch.from_numpy(syn.synth_tensor(f"bench_ctx_s{rank}", (n_ctx, D), "uniform", 0.6)).cuda()
        gl = torch.from_numpy(syn.synth_tensor(f"bench_ctx_g{rank}", (n_ctx, D), "uniform", 0.6)).cuda()
        hs, hg = expand_contexts(st, gl, B // n_ctx)

        def one_plan():
            x, chain = dm.plan({0: hs, 63: hg}, B, 64, n0, None, extra, return_chain=True, n_per_context=B // n_ctx, **guide_kw)
            if world > 1:   # the path's one exchange step: all-gather of the planned trajectories (RCCL over xGMI), timed
                gathered = gather_trajectories(x, n_ctx * world, B // n_ctx)
                assert gathered.shape[0] == world * B
            return chain

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        chain = one_plan()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        chain = one_plan()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert chain.shape == (T + n0 + 1, B, 64, D) and bool(torch.isfinite(chain[-1]).all())

    steps_per_plan = T + n0
    value = world * args.steps * steps_per_plan / dt
    out = {
        "metric": "denoising-steps/s", "value": round(value, 2), "unit": "denoising-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {env_id}-{robot} shape, {B} trajectories ({n_ctx} context(s)) x H=64 x D={D}, T={T} (+{n0}) "
                               f"reverse steps, {'guided (collision + GP prior, 5 guide steps on the last T/4+5 iterations)' if guided else 'unguided'}, "
                               f"U-Net dim_mults {mults}, one plan = one bench step",
                   "parallelism": ("single" if world == 1 else "replicas (no collective)" if n_ctx == 1
                                   else "contexts sharded over ranks, one all_gather of the final trajectories per plan"),
                   "denoising_steps_per_plan": steps_per_plan,
                   "trajectory_steps_per_s": round(value * B, 1)},
        "plan_wall_clock_ms": round(dt / args.steps * 1e3, 3),
    }
    if rank == 0 and not args.no_roofline:
        roof, table, unet_flops = roofline_leg(dm, B, T)
        out["roofline"] = roof
        # whole-plan view: algorithmic bytes (weights once per step + 4 tensor passes) and FLOPs
        w_bytes = sum(int(v.numel()) for v in sd.values()) * 4
        bytes_step = w_bytes + 4 * (B * 64 * D * 4)
        plan_s = dt / args.steps
        out["plan_roofline"] = {
            "algorithmic_bytes_per_step": bytes_step, "hbm_GBps": round(bytes_step * steps_per_plan / plan_s / 1e9, 2),
            "hbm_frac": round(bytes_step * steps_per_plan / plan_s / 1e9 / HBM_PEAK_GBS, 5),
            "algorithmic_flops_per_step": unet_flops, "fp32_TFLOPs": round(unet_flops * steps_per_plan / plan_s / 1e12, 3),
            "fp32_peak_frac": round(unet_flops * steps_per_plan / plan_s / 1e12 / FP32_PEAK_TFLOPS, 4)}
        if os.environ.get("MPDX_BENCH_TABLE"):
            for k, ms, fl, nl in table:
                print(f"# {ms*1e3:9.1f} us  {nl:3d} launches  {fl/ (ms*1e-3)/1e12 if ms > 0 else 0:7.2f} TF/s  {k}", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(sd, D, T, B, n0)
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
