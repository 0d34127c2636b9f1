#!/usr/bin/env python3
"""bench.py - denoising-steps/s and plan wall-clock of the reverse-diffusion planning loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 one rank per GPU under torch.distributed.run
(a plain `python bench.py --gpus N` re-executes itself under it).  A "step" here is ONE FULL PLAN = the region the
reference times (scripts/inference/inference.py:248-258): initial noise, all T+5 reverse updates of the whole batch,
chain materialisation.
  metric  = denoising-steps/s = (T+5) * plans / wall ; ms_per_step = plan wall-clock in ms.
Workload (BASELINE.json configs[1]): EnvDense2D-RobotPointMass shape - 100 trajectories x H=64 x D=4 (pos+vel),
T=100 diffusion steps (+5 without noise), U-Net dim_mults (1,2,4,8), unguided, fp32, synthetic formula-defined weights.
N>1: the single-context plan does not shard (B=100 does not fill a GPU): N independent replicas, no collective in the
data path (SURVEY.md 8e "replicas only"); value is the aggregate over ranks.  The path that DOES shard (BASELINE
configs[4]: independent start/goal contexts, one all-gather at the end) is reported beside it in the `sharded` sub-record.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def build_guide(env_id, robot, T, device):
    """The guide exactly as inference.py:188-236 builds it (weights 1e-2 / 1e-7, 5 guide steps, last quarter of the loop)."""
    import torch
    import mpd_public_amd as m
    from math import ceil
    ds = m.TrajectoryDataset(env_id, robot, tensor_args={"device": torch.device(device), "dtype": torch.float32})
    H_, dt_ = 64, 5.0 / 64
    cl = [m.CostCollision(ds.robot, H_, field=f, sigma_coll=1.0) for f in ds.task.get_collision_fields()]
    wl = [1e-2] * len(cl)
    cl.append(m.CostGPTrajectory(ds.robot, H_, dt_, sigma_gp=1.0)); wl.append(1e-7)
    guide = m.GuideManagerTrajectoriesWithVelocity(ds, m.CostComposite(ds.robot, H_, cl, weights_cost_l=wl), clip_grad=True,
                                                   interpolate_trajectories_for_collision=True).to(device)
    return dict(guide=guide, n_guide_steps=5, t_start_guide=ceil(0.25 * T))


# ------------------------------------------------------------------------------------------------------ roofline leg
def _unit_classes(lib, hdl, B, names, flops, n):
    """launch unit -> (class key, kernel name as rocprofv3 prints it)."""
    buf = C.create_string_buffer(64)
    out = []
    for i in range(n):
        nm = names[i].decode()
        li = lib.mpdx_unet_unit_layer(hdl, B, i)
        if li >= 0:
            lib.mpdx_unet_layer_tile(hdl, li, B, buf, 64)
            tile = buf.value.decode()
            kind = "conv_k5_gn_mish" if ".block.0." in nm else ("conv_k1" if "residual" in nm else ("down_k3s2" if "downs" in nm else "up_k4s2"))
            ws = tile.split(" ")[0] if tile.startswith("ws") else ""   # weight-stationary persistent kernels (large batches): ws (K split over
            # the waves, conv_ws.hpp), wsn (whole K per wave) / wsp (a pair of waves per tile) (conv_wsn.hpp)
            # kernel name as rocprofv3 prints it, down to the tile arguments (the PMC traffic files are keyed by full kernel names)
            mtnt = tile.split(" ")[-1].split("/")[0].split("x")
            if lib.mpdx_unet_unit_is_pair(hdl, B, i):
                kn = "conv_wsp_kernel" if ws == "wsp" else "conv_ws_kernel<32, 16, true" if ws else f"conv_pair_kernel<{mtnt[0]}, {mtnt[1]}"
                out.append((f"conv_k5_gn_mish+conv_k1 pair[{tile}] {flops[i]:.3e} flop", kn))
            else:
                mode = {"conv_k5_gn_mish": "0, 5, 1", "conv_k1": "0, 1, 0", "down_k3s2": "1, 3, 0", "up_k4s2": "2, 4, 0"}[kind]
                kn = (f"conv_wsn_kernel<{'2' if kind == 'up_k4s2' else '0'}, " if ws == "wsn" else "conv_ws_kernel<16, 32, false" if ws
                      else f"conv_block_kernel<{mode}, {mtnt[0]}, {mtnt[1]}, ")
                out.append((f"{kind}[{tile}] {flops[i]:.3e} flop", kn))
        elif nm.startswith("fused"):
            out.append(("fused level programs (fused_program_kernel<...>: whole-trajectory U-Net levels)", "fused_program_kernel"))
        else:
            out.append((nm, "final_step_kernel"))
    return out


_NOT_PLANNING = ("train", "k_train", "k_fused_train", "planner", "k_planner", "loss", "fused_bwd")


def csrc_fingerprint():
    """sha256 (16 hex digits) over the kernel sources (mpd_public_amd/csrc/*): a PMC traffic file carries the fingerprint of the sources it
    was measured on (tools/pmc_traffic.py), bench.py compares it with the sources it runs - a stale file is visible without git (the
    GPU box gets a snapshot without .git)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "mpd_public_amd" / "csrc").glob("*")):
        if f.suffix in (".hpp", ".hip") and not f.name.startswith(_NOT_PLANNING):   # the planning path's kernels: the training / planner families have no PMC records
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _pmc_traffic(B, kernel, pattern="r*_pmc_traffic*.json"):
    """HBM-side bytes per launch of `kernel` at batch B from the newest profiles/r*_pmc_traffic*.json that covers (batch, kernel):
    PMC counters cannot be read from inside this process; they are collected by tools/r04_evidence.sh (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes, gfx950 x2 fetch correction, MI355X_MICROARCH.md section HBM) on this same command and
    committed under profiles/.  Returns (bytes or None, file name, age): age says whether the file was measured on the kernel sources
    this run uses (fingerprints) - a stale file stays usable as an indication, and is labelled."""
    import re

    def _round_key(f):   # r02_ (a round's final evidence) after r02a_, r02b_ (mid-round states), rounds ascending
        m = re.match(r"r(\d+)([a-z]?)_", f.name)
        return (int(m.group(1)), m.group(2) or "~") if m else (-1, "")
    pats = ("fused_program_kernel", "fused_level_kernel") if kernel.startswith("fused") else (kernel,)
    for f in sorted((ROOT / "profiles").glob(pattern), key=_round_key, reverse=True):
        try:
            rec = json.loads(f.read_text())
        except Exception:
            continue
        if rec.get("batch") != B:
            continue
        hit = [v for k, v in rec.get("kernels", {}).items() if any(p in k for p in pats)]
        if not hit and "<" in kernel:   # an older file of a build whose kernels had other template arguments: match the family
            hit = [v for k, v in rec.get("kernels", {}).items() if kernel.split("<")[0] in k]
        if not hit:
            continue
        nl = sum(v["launches"] for v in hit)
        traffic = int(sum(v["traffic_bytes_per_launch"] * v["launches"] for v in hit) / nl)
        fp = rec.get("csrc_fingerprint")
        age = {"profile_sources": fp, "current_sources": csrc_fingerprint(),
               "measured_on_these_sources": (fp == csrc_fingerprint()) if fp else None}
        return traffic, f.name, age
    return None, None, None


def _rocprof_class_us(B, kernel, launches_per_pass):
    """Per-pass kernel time of a launch class from the newest committed rocprofv3 summary (profiles/r*_kernel_stats.json, written by
    tools/kernel_stats_json.py from `rocprofv3 --kernel-trace --stats` of this command) that was measured at batch B ON THE KERNEL SOURCES
    THIS RUN USES (fingerprints equal) - else (None, None).  = launches_per_pass x the call-weighted average duration of the class's kernels."""
    import re

    def _round_key(f):
        m = re.match(r"r(\d+)([a-z]?)_", f.name)
        return (int(m.group(1)), m.group(2) or "~") if m else (-1, "")
    pats = ("fused_program_kernel", "fused_level_kernel") if kernel.startswith("fused") else (kernel,)
    for f in sorted((ROOT / "profiles").glob("r*_kernel_stats.json"), key=_round_key, reverse=True):
        try:
            rec = json.loads(f.read_text())
        except Exception:
            continue
        if rec.get("batch") != B or rec.get("csrc_fingerprint") != csrc_fingerprint():
            continue
        hit = [v for k, v in rec.get("kernels", {}).items() if any(p in k for p in pats)]
        calls = sum(v["calls"] for v in hit)
        if not calls:
            continue
        return launches_per_pass * sum(v["calls"] * v["avg_ns"] for v in hit) / calls * 1e-3, f.name
    return None, None


def _alg_bytes(c):
    """algorithmic bytes of one launch of a class: its weights once + its input and output activations once.  Only the FLOP count is known
    per class here, so this is derived for the record from it where the shape is regular: None otherwise (DESIGN.md section 3 gives the
    per-kernel figures)."""
    return c.get("algorithmic_bytes_per_launch")


def _unet_pass_flops(dm, B, T):
    """algorithmic FLOPs of one U-Net pass at batch B from the library's own layer table (mpdx_unet_profile runs one pass)"""
    import torch
    from mpd_public_amd import _lib
    lib = _lib.load()
    hdl, packed, tab, wsb = dm.model.engine(T, B)
    x = torch.zeros(B, 64, dm.state_dim, device="cuda")
    cap = 128
    ms_ = (C.c_float * cap)(); fl_ = (C.c_double * cap)(); nm_ = (C.c_char_p * cap)(); n_ = C.c_int()
    _lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), 0, B, wsb.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream, cap, ms_, fl_, nm_, C.byref(n_)), "mpdx_unet_profile")
    return float(sum(fl_[k] for k in range(n_.value)))


def _train_launches_from_profile(B):
    """launches per training iteration from the newest committed rocprofv3 summary of the training loop at batch B (profiles/r*_train*_kernel_stats.csv,
    tools/r06_evidence.sh): sum of Calls / the iteration count (= Calls of adam_kernel, launched once per iteration)."""
    import csv
    import re
    name = "train_kernel_stats.csv" if B == 32 else f"train{B}_kernel_stats.csv"

    def _round_key(f):
        m = re.match(r"r(\d+)([a-z]?)_", f.name)
        return (int(m.group(1)), m.group(2) or "~") if m else (-1, "")
    for f in sorted((ROOT / "profiles").glob(f"r*_{name}"), key=_round_key, reverse=True):
        try:
            rows = list(csv.DictReader(open(f)))
            iters = sum(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
            if not iters:
                continue
            mine = sum(int(r["Calls"]) for r in rows if "mpdx::" in r["Name"])
            return {"file": f"profiles/{f.name}", "launches_per_iteration": round(mine / iters, 1), "iterations_profiled": iters,
                    "what": "native (mpdx::) kernel launches per iteration, warm-up iterations included"}
        except Exception:
            continue
    return None


def roofline_leg(dm, B, T, reps=30):
    """Where the time of one U-Net pass (= one denoising step, unguided) goes, per launch class, measured IN SITU: for every
    maximal run of consecutive launches of one class, one HIP-event pair on the launch stream brackets that run inside
    `reps` real passes (mpdx_unet_time_units) - real predecessors, cold weights, event cost amortised over the run.
    The `roofline` object describes the class with the LARGEST share of the pass; `classes` lists all of them."""
    import torch
    from mpd_public_amd import _lib
    lib = _lib.load()
    hdl, packed, tab, ws = dm.model.engine(T, B)
    x = torch.randn(B, 64, dm.state_dim, device="cuda")
    cap = 128
    ms = (C.c_float * cap)()
    fl = (C.c_double * cap)()
    names = (C.c_char_p * cap)()
    n = C.c_int()
    st = torch.cuda.current_stream().cuda_stream
    tt = T // 2
    _lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), tt, B, ws.data_ptr(), st,
                                     cap, ms, fl, names, C.byref(n)), "mpdx_unet_profile")
    n = n.value
    cls = _unit_classes(lib, hdl, B, names, fl, n)
    n_units = n if lib.mpdx_unet_unit_layer(hdl, B, n - 1) >= 0 or names[n - 1].decode().startswith("fused") else n - 1  # final kernel is not a unit
    runs, i = [], 0
    while i < n_units:
        j = i
        while j + 1 < n_units and cls[j + 1][0] == cls[i][0]:
            j += 1
        runs.append((i, j))
        i = j + 1
    out = C.c_float()
    table = OrderedDict()
    whole = C.c_float()
    _lib.check(lib.mpdx_unet_time_units(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), tt, B, ws.data_ptr(), st,
                                        0, n_units - 1, reps, C.byref(whole)), "mpdx_unet_time_units")
    for (a, b) in runs:
        _lib.check(lib.mpdx_unet_time_units(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), tt, B, ws.data_ptr(), st,
                                            a, b, reps, C.byref(out)), "mpdx_unet_time_units")
        e = table.setdefault(cls[a][0], {"us": 0.0, "flop": 0.0, "launches": 0, "kernel": cls[a][1], "longest_run": 0, "bytes": 0.0})
        e["bytes"] += sum(float(lib.mpdx_unet_unit_bytes(hdl, B, k)) for k in range(a, b + 1))
        e["us"] += out.value * 1e3
        e["flop"] += sum(fl[k] for k in range(a, b + 1))
        e["launches"] += b - a + 1
        e["longest_run"] = max(e["longest_run"], b - a + 1)
    # DIFFERENTIAL durations (the headline): a class costs the pass what the pass loses when the class's launches are left out - `dreps` back-to-back
    # passes between ONE event pair, with everything and without the class (mpdx_unet_time_without).  No event sits next to the measured launches: the
    # bracketed figure above carries ~5 us per event pair (marker processing + a dispatch gap the un-instrumented stream does not have): 15 % of a 35-us launch.
    dreps = max(reps, 60)
    full_pass = C.c_float()
    diff_ok = n_units <= 64

    def _pass_without(mask):
        v = C.c_float()
        _lib.check(lib.mpdx_unet_time_without(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x.data_ptr(), tt, B, ws.data_ptr(), st,
                                              C.c_uint64(mask), dreps, C.byref(v)), "mpdx_unet_time_without")
        return v.value * 1e3
    if diff_ok:
        full_us = min(_pass_without(0), _pass_without(0))
        for key, e in table.items():
            mask = 0
            for (a, b) in runs:
                if cls[a][0] == key:
                    for k in range(a, b + 1):
                        mask |= 1 << k
            e["us_diff"] = max(full_us - min(_pass_without(mask), _pass_without(mask)), 0.0)
    tot_us = sum(e["us"] for e in table.values())
    classes = []
    for k, e in sorted(table.items(), key=lambda kv: -kv[1]["us"]):
        us = e.get("us_diff") or e["us"]   # the differential duration when it was measured, else the bracketed one
        tf = e["flop"] / (us * 1e-6) / 1e12 if us > 0 else 0.0
        tfb = e["flop"] / (e["us"] * 1e-6) / 1e12 if e["us"] > 0 else 0.0
        classes.append({"class": k, "kernel": e["kernel"], "launches_per_pass": e["launches"], "us_per_pass": round(us, 2),
                        "share": round(e["us"] / tot_us, 4), "avg_launch_us": round(us / e["launches"], 2),
                        "flop_per_pass": e["flop"], "tflops": round(tf, 2), "frac": round(tf / FP32_PEAK_TFLOPS, 4),
                        "bracketed": {"us_per_pass": round(e["us"], 2), "avg_launch_us": round(e["us"] / e["launches"], 2), "frac": round(tfb / FP32_PEAK_TFLOPS, 4)},
                        "duration": "differential" if e.get("us_diff") else "bracketed",
                        "algorithmic_bytes_per_launch": int(e["bytes"] / e["launches"])})
    dom = dict(classes[0])
    dom["kernel_pattern"] = dom["kernel"]
    # HBM-side traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process; they are
    # collected by tools/r02_evidence.sh (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, gfx950 x2 fetch
    # correction, MI355X_MICROARCH.md section HBM) on this same command and committed under profiles/.
    traffic, traffic_src, traffic_age = _pmc_traffic(B, dom["kernel_pattern"])
    for c in classes:   # every class: HBM-side traffic per launch of ITS kernel at THIS batch (null when no PMC pass covers it)
        tr, src, _age = _pmc_traffic(B, c["kernel"])
        c["traffic"] = tr
        c["traffic_source"] = src
        c["traffic_over_algorithmic"] = round(tr / c["algorithmic_bytes_per_launch"], 2) if tr and c["algorithmic_bytes_per_launch"] else None
    # The HEADLINE figure is measured LIVE in this run (in-situ HIP events on the launch stream, contract section 4; ADVICE r5: a committed profile's number
    # ignores this box, its clocks and the runtime switches).  The committed rocprofv3 --kernel-trace --stats summary of the same command - when it was
    # measured on these kernel sources (fingerprint) - is reported BESIDE it (`rocprof`): the two must agree (the in-situ figure carries the event pair and
    # the gaps inside a run of launches, ~8 % at cfg 2).
    rp_us, rp_file = _rocprof_class_us(B, dom["kernel_pattern"], dom["launches_per_pass"])
    rocprof = None
    if rp_us:
        rp_tf = dom["flop_per_pass"] / (rp_us * 1e-6) / 1e12
        rocprof = {"achieved": round(rp_tf, 2), "frac": round(rp_tf / FP32_PEAK_TFLOPS, 4), "avg_launch_us": round(rp_us / dom["launches_per_pass"], 2),
                   "file": f"profiles/{rp_file}", "in_situ_over_rocprof": round((dom["us_per_pass"] / rp_us), 3),
                   "note": "rocprofv3 --kernel-trace --stats average of the class's kernels on these kernel sources (another box of the pool): auxiliary"}
    roof = {"bound": "mfma", "achieved": dom["tflops"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
            "duration_source": ("in situ, differential: (U-Net pass) - (U-Net pass without the class's launches), each = back-to-back passes between one HIP-event "
                                "pair on the launch stream of THIS run" if dom.get("duration") == "differential" else
                                "in situ: one HIP-event pair around each run of the class's launches inside real U-Net passes of THIS run"),
            "frac_in_situ": dom["frac"], "frac_bracketed": dom["bracketed"]["frac"], "frac_rocprof": (rocprof or {}).get("frac"), "rocprof": rocprof,
            "bracketed": dom["bracketed"],
            "traffic": traffic, "traffic_source": traffic_src, "traffic_age": traffic_age,
            "traffic_over_algorithmic": (round(traffic / _alg_bytes(dom), 2) if traffic and _alg_bytes(dom) else None),
            "kernel": dom["kernel"].split("<")[0], "kernel_instance": dom["kernel"], "class": dom["class"],
            "launches_per_unet_pass": dom["launches_per_pass"], "avg_launch_us": dom["avg_launch_us"], "share_of_pass": dom["share"],
            "algorithmic_flop_per_launch": dom["flop_per_pass"] / dom["launches_per_pass"],
            "timed": f"differential: {dreps} back-to-back real U-Net passes between one HIP-event pair, with and without the class (min of 2 each); bracketed: one "
                     f"HIP-event pair per run of consecutive launches of the class inside {reps} real passes",
            "unet_pass_us": round(whole.value * 1e3, 1), "unet_pass_us_sum_of_classes": round(tot_us, 1),
            "unet_pass_tflops": round(sum(fl[k] for k in range(n)) / (whole.value * 1e-3) / 1e12, 3),
            "classes": classes}
    if dom["kernel_pattern"].startswith("fused"):
        # one workgroup = one trajectory: a batch of B trajectories occupies min(B, 256) of the 256 CUs, so the fp32 MFMA rate these
        # programs can reach is that fraction of the chip peak (`frac` above stays achieved / CHIP peak)
        cus = min(B, 256)
        roof["occupancy"] = {"workgroups": B, "cus_in_use": cus, "peak_at_occupancy_tflops": round(FP32_PEAK_TFLOPS * cus / 256, 1),
                             "frac_of_peak_at_occupancy": round(dom["tflops"] / (FP32_PEAK_TFLOPS * cus / 256), 4)}
    return roof, sum(fl[k] for k in range(n))


def cpu_baseline_leg(sd, D, T, B, n0, min_seconds=10.0, max_plans=12):
    """The CPU oracle (oracle/, a port of the reference's algorithm validated against it) timed on this box's host
    cores on the SAME workload: whole plans until >= min_seconds of CPU work (bounded sample).

    Two forms are measured and the faster is the record's `value`:
      * one process (how the reference runs: one Python process, ATen's intra-op threads) at the best thread count of a probe;
      * the batch split over several processes (trajectories are independent: slices of the batch, the same noise rows, identical results) -
        what a user with this host would do to fill its cores, the per-layer ATen ops of a 100-trajectory batch not scaling past 8-32 threads."""
    import torch
    from oracle import diffusion as odiff
    from mpd_public_amd import synthetic as syn
    # thread count: torch's default (all logical cores) oversubscribes these small per-layer ops badly on big hosts
    # (measured: 128 threads -> 5.9 steps/s, slower than 8 threads); probe a few settings on 2 steps and keep the best.
    host_cores = os.cpu_count() or 8
    probe_noise = torch.randn((4, B, 64, D))
    hc0 = {0: torch.zeros(D), 63: torch.zeros(D)}
    best = (1e30, torch.get_num_threads())
    for nthr in sorted({8, 16, 32, 64, min(host_cores, 128)}):
        if nthr > host_cores:
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
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    single = {"value": round(plans * (T + n0) / dt, 2), "cores": cores, "plan_wall_s": round(dt / plans, 3),
              "sample": f"{plans} full plan(s) of {T + n0} steps, B={B}, torch-CPU fp32 oracle, ONE process on {cores} threads (best of a probe over "
                        f"8..{min(host_cores, 128)} on this {host_cores}-logical-core host), {dt:.1f} s"}
    rec = {"value": single["value"], "unit": "denoising-steps/s", "cores": cores, "host_cores": host_cores, "host_cpu": cpu_model, "kind": "port",
           "sample": single["sample"],
           "note": f"a step is ~1 400 small ATen calls (46 convolutions of <= 100 x 64 positions, GroupNorm, Mish): the path is dispatch-bound on a CPU (a plan = "
                   "~147 k calls) and its intra-op parallelism saturates at 8-32 threads (more threads run SLOWER: probe); `multi_process` splits the batch over "
                   "processes instead - every slice still pays every call, so it only wins on hosts where one process is compute-bound",
           "plan_wall_s": single["plan_wall_s"], "single_process": single}
    try:
        multi = _cpu_baseline_multiprocess(sd, hc, noise, D, T, B, n0, host_cores, single["plan_wall_s"])
    except Exception as e:   # the single-process figure stands
        multi = {"error": f"{type(e).__name__}: {e}"}
    rec["multi_process"] = multi
    if multi.get("value", 0.0) > rec["value"]:
        rec.update(value=multi["value"], cores=multi["cores"], sample=multi["sample"], plan_wall_s=multi["plan_wall_s"])
    return rec


def _cpu_baseline_worker(idx, cmd_q, res_q, sd, hc, noise, T, n0):
    """one process of the multi-process CPU baseline: plans its slice of the batch when told to (cpu_baseline_leg only)"""
    import torch
    from oracle import diffusion as odiff
    try:
        while True:
            cmd = cmd_q.get()
            if cmd is None:
                return
            b0, b1, nthr, plans = cmd
            torch.set_num_threads(nthr)
            nz = noise[:, b0:b1].contiguous()
            odiff.run_inference(sd, hc, nz[:3], 1, n_diffusion_steps_without_noise=1, noise_std=0.5)   # warm-up at this slice size / thread count
            res_q.put((idx, "ready", 0.0))
            cmd_q.get()   # "go"
            t0 = time.perf_counter()
            for _ in range(plans):
                odiff.run_inference(sd, hc, nz, T, n_diffusion_steps_without_noise=n0, noise_std=0.5)
            res_q.put((idx, "done", time.perf_counter() - t0))
    except Exception as e:   # noqa: BLE001 - reported to the parent, which falls back to the single-process figure
        res_q.put((idx, "error", f"{type(e).__name__}: {e}"))


def _cpu_baseline_multiprocess(sd, hc, noise, D, T, B, n0, host_cores, single_plan_s, budget_s=10.0):
    """the batch split over P processes x nthr threads (spawned once; a few (P, nthr) settings are tried on one plan each, the best one is
    then timed on >= 3 plans).  Wall-clock = from the common start signal to the LAST process's finish."""
    import multiprocessing as mp
    phys = max(1, host_cores // 2)   # SMT siblings do not help these ATen loops
    cands = []
    # (measured on the pool's 256-logical-core hosts, six settings from 2 x 16 to 16 x 8: the best, 4 x 8, took 2.05 s per plan against 1.40 s for ONE
    #  process - a plan is ~147 k ATen calls of ~9.5 us, dispatch-bound, and every slice pays all of them; three settings are kept as the probe)
    for P, nthr in ((4, 8), (8, 8), (2, 16)):
        nthr = max(1, min(nthr, phys // P))
        if P <= B and P * 2 <= host_cores and (P, nthr) not in cands:
            cands.append((P, nthr))
    if not cands:
        return {"skipped": f"{host_cores} logical cores: nothing to split over"}
    Pmax = max(p for p, _ in cands)
    ctx = mp.get_context("spawn")   # (a forked child of a process that holds a HIP context is not safe)
    res_q = ctx.Queue()
    cmd_qs = [ctx.Queue() for _ in range(Pmax)]
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    procs = [ctx.Process(target=_cpu_baseline_worker, args=(i, cmd_qs[i], res_q, sd_cpu, hc, noise, T, n0), daemon=True) for i in range(Pmax)]
    for p in procs:
        p.start()

    import queue as _queue

    def collect(P, want, timeout):
        t_end = time.perf_counter() + timeout
        for _ in range(P):
            while True:
                try:
                    _, tag, val = res_q.get(timeout=1.0)
                    break
                except _queue.Empty:
                    if any(not p.is_alive() for p in procs[:P]):
                        raise RuntimeError("a worker process died (spawn failed?)") from None
                    if time.perf_counter() > t_end:
                        raise RuntimeError(f"no answer within {timeout:.0f} s") from None
            if tag != want:
                raise RuntimeError(f"worker: {val}")

    def run(P, nthr, plans, timeout):
        bounds = [(B * i) // P for i in range(P + 1)]
        for i in range(P):
            cmd_qs[i].put((bounds[i], bounds[i + 1], nthr, plans))
        collect(P, "ready", timeout)
        t0 = time.perf_counter()
        for i in range(P):
            cmd_qs[i].put("go")
        collect(P, "done", timeout)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    try:
        tried, best = [], None
        for (P, nthr) in cands:
            if time.perf_counter() - t_begin > budget_s:
                break
            w = run(P, nthr, 1, timeout=120.0 + 4 * single_plan_s)
            tried.append({"processes": P, "threads_each": nthr, "plan_wall_s": round(w, 3)})
            if best is None or w < best[0]:
                best = (w, P, nthr)
        w1, P, nthr = best
        plans = int(max(2, min(12, 5.0 / max(w1, 1e-3))))
        w = run(P, nthr, plans, timeout=120.0 + 4 * plans * w1)
    finally:
        for q in cmd_qs:
            q.put(None)
        for p in procs:
            p.join(timeout=5.0)
            if p.is_alive():
                p.terminate()
    return {"value": round(plans * (T + n0) / w, 2), "cores": P * nthr, "processes": P, "threads_each": nthr, "plan_wall_s": round(w / plans, 3),
            "probe": tried,
            "sample": f"{plans} full plan(s) of {T + n0} steps, B={B} split over {P} processes x {nthr} threads (torch-CPU fp32 oracle, slices of the batch; "
                      f"best of {len(tried)} settings), {w:.1f} s"}


# ------------------------------------------------------------------------------------------------------ sub-records
def guide_flop_estimate(ds, guide_mgr, B, H=64):
    """FLOP ESTIMATE of one guide launch (DESIGN.md section 6 states the per-term counts): interpolation 64 -> N points, per point and
    collision field the SDF scan over its primitives (sphere 3d+3, box 6d+6, workspace face 2 flops), hinge + gradient (4d+6), Panda:
    forward kinematics (7 joint transforms ~ 70 flops each + 11 link spheres ~ 15 each) per point and a Jacobian-transpose row
    (7 joints x 14) per sphere and field; GP prior 30 q per support point; gather / clip / weight 12 D per support point and term."""
    N = guide_mgr.num_interpolated_points_for_collision if guide_mgr.interpolate_trajectories_for_collision else H
    D, q, d = ds.state_dim, ds.robot.q_dim, ds.env.dim
    panda = ds.robot.name == "RobotPanda"
    n_sph = 11 if panda else 1
    per_point = (490 + 11 * 15) if panda else 0.0
    n_terms = 0
    for c in guide_mgr.cost.cost_l:
        fld = getattr(c, "field", None)
        if fld is None:
            continue
        n_terms += 1
        if fld.objects is not None:
            ns, nb = len(fld.objects.sphere_radii), len(fld.objects.box_centers)
            per_point += n_sph * (ns * (3 * d + 3) + nb * (6 * d + 6) + 4 * d + 6)
        elif fld.ws_min is not None:
            per_point += n_sph * (2 * d * 2 + 4 * d + 6)
        else:   # self collision: 12 sphere pairs
            per_point += 12 * (3 * d + 3 + 4 * d + 6)
        if panda:
            per_point += n_sph * 7 * 14
    per_traj = N * (per_point + 3 * D) + H * (30 * q + 12 * D * (n_terms + 1))
    return float(B) * per_traj


def guided_leg(device, plans=5, reps=200):
    """BASELINE configs[2] and [3] - the GUIDED sampler north_star names (inference.py:188-258 is the timed region): plan wall-clock and
    denoising-steps/s, the guide kernel's time per launch (mpdx_guide_time: `reps` back-to-back launches between one HIP-event pair on
    the launch stream) against its algorithmic bytes (2 * B*H*D*4 + tables) and a FLOP estimate, and - checker leg - the
    plan figures of an <= 8-trajectory slice planned with the SAME noise by the HIP path and by the CPU oracle (fp32 and fp64): _guided_oracle_check."""
    import torch
    from mpd_public_amd import _lib, synthetic as syn
    lib = _lib.load()
    out = {}
    for cfg in ("cfg3", "cfg4"):
        env_id, robot, D, mults, T, B, n0, _, _ = CONFIGS[cfg]
        dm, sd = build_model(D, mults, T, device)
        dm.manual_seed(30)
        gk = build_guide(env_id, robot, T, device)
        g = gk["guide"]
        hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform", 0.6)).cuda(),
              63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform", 0.6)).cuda()}

        def one():
            return dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=True, n_diffusion_steps_without_noise=n0,
                                    noise_std_extra_schedule_fn=lambda t: 0.5, **gk)
        chain = one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(plans):
            chain = one()
        torch.cuda.synchronize()
        plan_s = (time.perf_counter() - t0) / plans
        n_guided = sum(1 for i in range(-n0, T) if i < gk["t_start_guide"])
        launches = n_guided * gk["n_guide_steps"]
        # guide kernel alone, on the planned trajectories
        x = chain[-1].contiguous()
        gp = g.device_params(x.device)
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(lib.mpdx_absmax(x.data_ptr(), flag.data_ptr(), B, B, 64, D, _lib.current_stream()), "mpdx_absmax")
        scratch = torch.empty_like(x)
        ms = C.c_float()
        _lib.check(lib.mpdx_guide_time(C.byref(gp), x.data_ptr(), scratch.data_ptr(), flag.data_ptr(), B, B, 64, D, reps,
                                       _lib.current_stream(), C.byref(ms)), "mpdx_guide_time")
        us = ms.value * 1e3
        table_bytes = int(gp.n_prim_floats) * 4
        abytes = 2 * B * 64 * D * 4 + table_bytes
        flop = guide_flop_estimate(g.dataset, g, B)
        gname = "guide_step_panda_kernel" if robot == "RobotPanda" else "guide_step_kernel"
        g_traffic, g_src, g_age = _pmc_traffic(B, gname, "r*_guidepmc_*.json")   # (own file family: the headline's lookup must not meet the D = 14 network's kernels)
        # rocprofv3 --pmc passes of `bench.py --config cfg3 / cfg4` (tools/r04_evidence.sh), None without a file
        rec = {"workload": f"{cfg}: {env_id}-{robot} shape, {B} trajectories x H=64 x D={D}, T={T} (+{n0}), guided: {n_guided} guided steps x "
                           f"{gk['n_guide_steps']} guide iterations = {launches} guide launches per plan",
               "plan_ms": round(plan_s * 1e3, 3), "denoising_steps_per_s": round((T + n0) / plan_s, 1),
               "guide_kernel": {"kernel": "guide_step_panda_kernel" if robot == "RobotPanda" else "guide_step_kernel",
                                "us_per_launch": round(us, 2), "launches_per_plan": launches,
                                "share_of_plan": round(us * 1e-6 * launches / plan_s, 4),
                                "timed": f"{reps} back-to-back launches between one HIP-event pair on the launch stream (mpdx_guide_time)",
                                "roofline": {"bound": "hbm", "achieved": round(abytes / (us * 1e-6) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(abytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                             "algorithmic_bytes_per_launch": abytes, "table_bytes": table_bytes, "traffic": g_traffic,
                                             "traffic_source": g_src, "traffic_age": g_age,
                                             "note": "latency-bound by construction: 2 * B*H*D*4 bytes + the primitive table per launch; FK / SDF "
                                                     "arithmetic on one workgroup per trajectory"},
                                "flop_estimate_per_launch": flop, "GFLOPs_estimate": round(flop / (us * 1e-6) / 1e9, 1),
                                "fp32_peak_frac_estimate": round(flop / (us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4)}}
        try:
            rec["oracle_check"] = _guided_oracle_check(dm, sd, g, gk, hc, D, T, n0, 4 if robot == "RobotPanda" else 8)
        except Exception as e:   # the record must survive a failing checker
            rec["oracle_check"] = {"error": f"{type(e).__name__}: {e}"}
        out[cfg] = rec
    return out


def _guided_oracle_check(dm, sd, g, gk, hc, D, T, n0, nb):
    """Checker leg: `nb` trajectories planned with the SAME pre-generated noise by the HIP path and by the CPU oracle in fp32 AND fp64
    (oracle/: U-Net, DDPM step and the autograd guide of oracle/costs.py).  The record is tests/helpers.py::guided_parity_record - the
    decidable form of north_star's "identical to 3 s.f." for a discontinuous (hinge / arg-min / norm-clip) chain: `equal_to_3sf` compares
    the HIP metrics kernel with an fp64 evaluation of the SAME plan over the waypoints whose fp64 hinge slack is not within 1e-5 of a
    margin (`ambiguous_waypoints` excluded), `chain_class` bounds the HIP chain's hinge flips against the fp64 chain by the fp32 CPU
    oracle's own (inference.py:288-297, 311-316)."""
    from helpers import guided_parity_record
    return guided_parity_record(dm, sd, g, gk, hc, T, n0, nb)


def train_small_model(root, env_id="EnvDense2D", robot_id="RobotPointMass", contexts=64, per_context=16, steps=3000, batch=32, lr=3e-4, T=25, opt=1):
    """The reference's generate_trajectories.py -> train.py on this GPU (baseline planners, native training step): a small trained model.
    -> (results dir = a model directory of the inference entry, record of the two stages)"""
    import contextlib
    import torch
    from mpd_public_amd import train as train_script
    from mpd_public_amd.generate_trajectories import generate_collision_free_trajectories as gen
    sub = f"{env_id}-{robot_id}"
    rec = {"model_id": sub, "contexts": contexts, "trajectories_per_context": per_context, "train_steps": steps, "batch": batch, "lr": lr,
           "n_diffusion_steps": T, "unet_dim_mults_option": opt}
    with contextlib.redirect_stdout(sys.stderr):   # (the stages print progress lines; the bench line is the only thing on stdout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_free = 0
        for ctx in range(contexts):
            d = os.path.join(root, "data_trajectories", sub, str(ctx))
            os.makedirs(d, exist_ok=True)
            _, nf = gen(env_id, robot_id, per_context, d, gpmp_opt_iters=300, seed=ctx)
            n_free += nf
        torch.cuda.synchronize()
        rec["generate_s"] = round(time.perf_counter() - t0, 2)
        rec["collision_free_training_trajectories"] = int(n_free)
        t0 = time.perf_counter()
        logs = os.path.join(root, "logs")
        _, _, losses = train_script.experiment(dataset_subdir=sub, data_dir=os.path.join(root, "data_trajectories"), results_dir=logs, n_diffusion_steps=T,
                                               unet_dim_mults_option=opt, batch_size=batch, lr=lr, num_train_steps=steps,
                                               steps_til_summary=max(50, steps // 10), steps_til_ckpt=steps, seed=1, summary_class=None, debug=False)
        torch.cuda.synchronize()
    rec["train_s"] = round(time.perf_counter() - t0, 2)
    vals = [float(v["diffusion_loss"]) for _, v in losses if "diffusion_loss" in v]
    rec["diffusion_loss_first_last"] = [round(vals[0], 5), round(vals[-1], 5)] if vals else None
    return logs, rec


def trained_leg(device, contexts=256, per_context=16, steps=20000, n_samples=100, n_contexts_planned=3, nb_check=8):
    """North_star's plan figures on TRAINED weights.  Formula-defined weights plan trajectories that all collide (collision-free rate 0.0 on both sides of
    every check); here the whole chain of the reference's scripts runs on this GPU first - generate_trajectories.py (RRT-Connect + GPMP2, f-4) -> train.py
    (the native training step, f-3) -> the EMA model plans `n_samples` trajectories per context unguided (`diffusion_prior`) and guided (`mpd`,
    inference.py:188-258) - and the guided plan of a slice is checked against the CPU oracle on the SAME trained weights and noise (guided_parity_record).
    Environment / limits are this package's synthetic stand-ins, so the figures are not the paper's; they are non-trivial."""
    import shutil
    import tempfile
    import torch
    import yaml
    import mpd_public_amd as m
    from math import ceil
    from mpd_public_amd.datasets import LimitsNormalizer
    env_id, robot_id, T, n0 = "EnvDense2D", "RobotPointMass", 25, 5
    root = tempfile.mkdtemp(prefix="mpdx_trained_")
    try:
        logs, rec = train_small_model(root, env_id, robot_id, contexts, per_context, steps, T=T)
        ta = {"device": torch.device(device), "dtype": torch.float32}
        ds = m.TrajectoryDataset(env_id=env_id, robot_id=robot_id, use_extra_objects=True, obstacle_cutoff_margin=0.05, include_velocity=True, tensor_args=ta)
        lm = yaml.safe_load(open(os.path.join(logs, "limits.yaml")))
        ds.normalizer = LimitsNormalizer(lm["mins"], lm["maxs"]).to(ta["device"])
        sd = torch.load(os.path.join(logs, "checkpoints", "ema_model_current_state_dict.pth"), map_location="cpu")
        net = m.TemporalUnet(n_support_points=64, state_dim=ds.state_dim, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[1])
        dm = m.GaussianDiffusionModel(model=net, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
        dm.load_state_dict(sd, strict=True)
        dm = dm.to(device).eval()
        dm.manual_seed(30)
        H_, dt_ = 64, 5.0 / 64
        cl = [m.CostCollision(ds.robot, H_, field=f, sigma_coll=1.0) for f in ds.task.get_collision_fields()]
        wl = [1e-2] * len(cl)
        cl.append(m.CostGPTrajectory(ds.robot, H_, dt_, sigma_gp=1.0)); wl.append(1e-7)
        guide = m.GuideManagerTrajectoriesWithVelocity(ds, m.CostComposite(ds.robot, H_, cl, weights_cost_l=wl), clip_grad=True,
                                                       interpolate_trajectories_for_collision=True).to(device)
        gk = dict(guide=guide, n_guide_steps=5, t_start_guide=ceil(0.25 * T))
        plans = {"diffusion_prior": [], "mpd": []}
        hc_first = None
        for c in range(n_contexts_planned):
            gen = torch.Generator(device=ta["device"]).manual_seed(30 + c)
            for _ in range(100):   # inference.py:161-169
                q = ds.task.random_coll_free_q(n_samples=2, device=ta["device"], generator=gen)
                if torch.linalg.norm(q[0] - q[1]) > ds.threshold_start_goal_pos:
                    break
            hc = ds.get_hard_conditions(torch.vstack((q[0], q[1])), normalize=True)
            hc_first = hc_first or hc
            for alg, kw in (("diffusion_prior", {}), ("mpd", gk)):
                dm.run_inference(None, hc, n_samples=n_samples, horizon=64, n_diffusion_steps_without_noise=n0, noise_std_extra_schedule_fn=lambda t: 0.5, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                x = dm.run_inference(None, hc, n_samples=n_samples, horizon=64, n_diffusion_steps_without_noise=n0, noise_std_extra_schedule_fn=lambda t: 0.5, **kw)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                xu = ds.unnormalize_trajectories(x)
                plans[alg].append({"context": c, "plan_ms": round(ms, 3), "collision_free_rate": round(float(ds.task.compute_fraction_free_trajs(xu)), 4),
                                   "collision_intensity": round(float(ds.task.compute_collision_intensity_trajs(xu)), 5)})
        rec["plans"] = plans
        rec["mean_collision_free_rate"] = {k: round(sum(r["collision_free_rate"] for r in v) / len(v), 4) for k, v in plans.items()}
        try:
            from helpers import guided_parity_record
            sd_unet = {k[len("model."):]: v.float() for k, v in sd.items() if k.startswith("model.")}   # the oracle's functional U-Net takes the bare keys
            chk = guided_parity_record(dm, sd_unet, guide, gk, hc_first, T, n0, nb_check)
            chk["weights"] = f"TRAINED here ({steps} iterations on {rec['collision_free_training_trajectories']} generated trajectories)"
            rec["oracle_check"] = chk
        except Exception as e:   # the record must survive a failing checker
            rec["oracle_check"] = {"error": f"{type(e).__name__}: {e}"}
        return rec
    finally:
        shutil.rmtree(root, ignore_errors=True)


def _all_ranks(dist, vals, device):
    """[world][len(vals)] float64 table of every rank's values (host-staged on gloo)."""
    import torch
    if dist is None:
        return [list(map(float, vals))]
    gloo = dist.get_backend() == "gloo"
    mine = torch.tensor(vals, dtype=torch.float64, device="cpu" if gloo else device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(v) for v in t.cpu()] for t in out]


class _Watchdog:
    """Arms a timer around a leg that ends in collectives: if the leg has not returned after `seconds`, rank 0 prints the record it
    holds so far (the headline line, with the leg marked as timed out) and the process exits with status 0 - a hung collective in a
    sub-record must not cost the driver its bench line."""

    def __init__(self, seconds, rank, out):
        import threading
        self.rank, self.out = rank, out
        self.t = threading.Timer(seconds, self._fire, args=(seconds,))
        self.t.daemon = True
        self.t.start()

    def _fire(self, seconds):
        if self.rank == 0:
            rec = dict(self.out)
            rec["sharded"] = {"error": f"watchdog: the sharded leg did not return within {seconds:g} s (a collective never completed)"}
            for k in ("cpu_baseline", "guided", "serving", "planner_baseline", "trained", "training"):
                rec.setdefault(k, None)
            emit(rec)
        os._exit(0)

    def cancel(self):
        self.t.cancel()


def _agree(dist, ok, device):
    """True iff EVERY rank passes ok=True (one MIN all-reduce of a flag): the ranks decide together whether to enter a step that only
    works if all of them do - a rank that already failed cannot leave its peers waiting in it."""
    import torch
    if dist is None:
        return bool(ok)
    gloo = dist.get_backend() == "gloo"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if gloo else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.cpu()[0]) == 1)


class _Stopwatch:
    """elapsed ms between marks: HIP events on the current stream for a GPU device, the host clock (after a synchronise) otherwise
    (the CPU rig of tests/test_parallel_cpu.py runs this leg's sharding / gather / record code on the gloo backend without a GPU)."""

    def __init__(self, device, n):
        import torch
        self.gpu = str(device).startswith("cuda")
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n)] if self.gpu else [0.0] * n

    def mark(self, i):
        if self.gpu:
            self.ev[i].record()
        else:
            self.ev[i] = time.perf_counter()

    def ms(self, i, j):
        return self.ev[i].elapsed_time(self.ev[j]) if self.gpu else (self.ev[j] - self.ev[i]) * 1e3


def sharded_leg(rank, world, dist, device, plans=2, n_ctx=None, n_samples=None, plan_fn=None, hop_timeout_s=60.0, hop_precheck=None):
    """BASELINE configs[4] per-GPU shard: 128 start/goal contexts x 50 Panda trajectories per rank (weak scaling: 128*N contexts in
    total), guided, per-trajectory hard conditions, per-context range tests, zero exchange during the loop, ONE gather of the
    planned trajectories at the end (RCCL over xGMI when N > 1).  Self-checking: the gathered tensor is verified block by block
    against checksums the owning ranks publish; both gather variants (all_gather_into_tensor / one-hop grouped send+recv over the
    direct links) are timed in the same run; per-rank plan times are reported as min / median / max, not only the max.
    n_ctx / n_samples / plan_fn: the CPU rig (tests/test_parallel_cpu.py: world 8 on gloo, one context per rank) replaces the planner by
    a stand-in `plan_fn(hs, hg) -> [B, 64, D]` so that the shard arithmetic, both gathers, the agreement protocol and the record's
    shape are exercised every round without a GPU; bench.py itself never passes them.  hop_precheck(rank) -> error string or None:
    a rank's own pre-flight check of the one-hop form (tests inject a failure there).
    A failure INSIDE a one-hop attempt (a peer dying between its sends) can leave the communicator unusable: main() therefore arms a
    watchdog (_Watchdog) around this leg at N > 1 that prints the headline line and exits if the leg does not return."""
    import statistics
    import torch
    from mpd_public_amd import synthetic as syn
    from mpd_public_amd.parallel import expand_contexts, gather_trajectories, verify_gather
    env_id, robot, D, mults, T, B, n0, _, cfg_ctx = CONFIGS["cfg5"]
    n_ctx = int(n_ctx or cfg_ctx)
    n_samples = int(n_samples or B // cfg_ctx)
    B = n_ctx * n_samples
    on_gpu = str(device).startswith("cuda")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if dist is not None and dist.get_world_size() != world:
        raise RuntimeError(f"process group has {dist.get_world_size()} ranks, the launcher announced {world}")
    st = torch.from_numpy(syn.synth_tensor(f"bench_ctx_s{rank}", (n_ctx, D), "uniform", 0.6)).to(device)
    gl = torch.from_numpy(syn.synth_tensor(f"bench_ctx_g{rank}", (n_ctx, D), "uniform", 0.6)).to(device)
    hs, hg = expand_contexts(st, gl, n_samples)
    if plan_fn is None:
        dm, _sd = build_model(D, mults, T, device)
        dm.manual_seed(1000 + rank)
        gk = build_guide(env_id, robot, T, device)

        def plan_fn(hs_, hg_):
            return dm.plan({0: hs_, 63: hg_}, B, 64, n0, None, lambda t: 0.5, return_chain=False, n_per_context=n_samples, **gk)[0]
        try:   # algorithmic FLOPs of one U-Net pass at this batch (the library's own layer table) -> the plan's fraction of the fp32 MFMA peak
            unet_flop_per_step = _unet_pass_flops(dm, B, T)
        except Exception:   # the record must survive a failing helper
            unet_flop_per_step = None
    else:
        unet_flop_per_step = None
    sw = _Stopwatch(device, 3)
    plan_ms, gather_ms, hop_ms = [], [], []
    verified, hop_error = True, None
    x = g = None
    for it in range(plans + 1):
        if dist is not None:
            dist.barrier()
        sync()
        sw.mark(0)
        x = plan_fn(hs, hg)
        sw.mark(1)
        g = gather_trajectories(x, n_ctx * world, n_samples, force_collective=dist is not None, mode="collective")
        sw.mark(2)
        sync()
        assert g.shape[0] == world * B and bool(torch.isfinite(g[-1]).all())
        if it == plans:   # transport check on the last plan: every block against its owner's checksum
            verified = verify_gather(g, x, n_ctx * world, n_samples)
        if it > 0:
            plan_ms.append(sw.ms(0, 1)); gather_ms.append(sw.ms(1, 2))
    # the default form's record is complete HERE (the headline comes from it alone).  The one-hop form is measured afterwards, on the last
    # plan's trajectories.  First contact with a real 8-GPU fabric happens in the driver's run, so the attempt is defensive: before EVERY
    # attempt the ranks agree (one MIN all-reduce) that none of them has failed yet - an asymmetric failure cannot strand the others in a
    # grouped send / receive - and every wait inside the attempt is bounded (a rank that dies mid-way makes its peers raise after the timeout).
    hop_ok = True
    hop_skipped_by_agreement = False
    if hop_precheck is not None:
        hop_error = hop_precheck(rank)
    for it in range(plans):
        g1 = None
        if not _agree(dist, hop_error is None, device):
            hop_skipped_by_agreement = hop_error is None
            hop_error = hop_error or "skipped: another rank reported a failure before this attempt"
            break
        try:
            if dist is not None:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            g1 = gather_trajectories(x, n_ctx * world, n_samples, force_collective=dist is not None, mode="one_hop", timeout_s=hop_timeout_s)
            sync()
            hop_ms.append((time.perf_counter() - t0) * 1e3)
        except Exception as e:
            hop_error = f"{type(e).__name__}: {e}"
        if hop_error is None and it == plans - 1 and g1 is not None:
            hop_ok = verify_gather(g1, x, n_ctx * world, n_samples) and bool(torch.equal(g, g1))
    if hop_error is not None or not hop_ms:
        hop_ms = [float("nan")]
    table = _all_ranks(dist, [statistics.median(plan_ms), max(plan_ms), statistics.median(gather_ms), max(gather_ms),
                              statistics.median(hop_ms), max(hop_ms), 1.0 if verified else 0.0, 0.0 if hop_error is None else 1.0,
                              1.0 if hop_ok else 0.0], device)
    if not all(r[6] == 1.0 for r in table):
        raise RuntimeError("gathered trajectories do not match the per-rank checksums")
    per_rank_plan = [r[0] for r in table]
    pm, gm = max(r[1] for r in table), max(r[3] for r in table)
    hop_failed = any(r[7] != 0.0 for r in table)
    hm = float("nan") if hop_failed else max(r[5] for r in table)
    if not hop_failed and not all(r[8] == 1.0 for r in table):
        hop_failed, hop_error = True, "one-hop result does not match the per-rank checksums / the collective's result"
    nccl_env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_")) and k not in ("NCCL_ASYNC_ERROR_HANDLING",)}
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if on_gpu else None
    except Exception as e:   # (a build without the binding must not cost the record)
        rccl_version = f"unavailable ({type(e).__name__})"
    return {"workload": f"cfg5 shard per rank: {n_ctx} contexts x {n_samples} = {B} Panda trajectories, T={T} (+{n0}), guided; {n_ctx * world} contexts in total",
            "ranks": world, "backend": (dist.get_backend() if dist is not None else None),
            "process_group_world_size": (dist.get_world_size() if dist is not None else None),
            "cuda_device_count": (torch.cuda.device_count() if on_gpu else 0),
            "ranks_per_gpu": (max(1, world // max(1, torch.cuda.device_count())) if dist is not None and on_gpu else 1),
            "rccl_version": rccl_version, "hip_version": getattr(torch.version, "hip", None),
            "nccl_env": nccl_env or None,
            "collective": "all_gather_into_tensor of the final trajectories" if dist is not None else None,
            "headline_gather": "all_gather_into_tensor (the default mode; the one-hop form is reported beside it, never mixed into the headline)",
            "plan_ms_per_rank": {"min": round(min(per_rank_plan), 2), "median": round(statistics.median(per_rank_plan), 2),
                                 "max": round(max(per_rank_plan), 2), "all": [round(v, 2) for v in per_rank_plan]},
            "plan_ms_per_rank_max": round(pm, 2), "all_gather_ms_max": round(gm, 3),
            "gather_ms": {"all_gather_into_tensor": {"median_over_ranks": round(statistics.median(r[2] for r in table), 3), "max": round(gm, 3)},
                          "one_hop_send_recv": ({"error": hop_error or "failed on another rank", "skipped_by_agreement": bool(hop_skipped_by_agreement)}
                                                if hop_failed else
                                                {"median_over_ranks": round(statistics.median(r[4] for r in table), 3), "max": round(hm, 3),
                                                 "timed": "host clock around the call, synchronised (after the main record); the ranks agree before "
                                                          "every attempt, waits bounded at %g s" % hop_timeout_s})},
            "gather_verified": "per-block position-weighted bit-pattern checksums published by the owning ranks match on every rank" +
                               ("" if hop_failed else "; both variants bit-identical"),
            "one_hop_denoising_steps_per_s": (None if hop_failed else round(world * (T + n0) / ((pm + hm) * 1e-3), 2)),
            "all_gather_bytes_per_rank": int(B * 64 * D * 4),
            "denoising_steps_per_s": round(world * (T + n0) / ((pm + gm) * 1e-3), 2),
            "unet_flop_per_step": unet_flop_per_step,
            "fp32_TFLOPs": (round(unet_flop_per_step * (T + n0) / (pm * 1e-3) / 1e12, 2) if unet_flop_per_step else None),
            "fp32_peak_frac": (round(unet_flop_per_step * (T + n0) / (pm * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if unet_flop_per_step else None),
            "fp32_peak_frac_note": "U-Net FLOPs only (the guide's are < 1 %) x (T + 5) / the slowest rank's plan time, per GPU",
            "trajectory_steps_per_s": round(world * B * (T + n0) / ((pm + gm) * 1e-3), 1), "scaling": "weak"}


def serving_leg(dm, D, T, n0, n_ctx=16, n=100, plans=3):
    """Several start/goal contexts BATCHED into one plan (per-trajectory hard conditions): the single-context chain is
    latency-bound and leaves most of the GPU idle, so a planning server batches queued requests."""
    import torch
    from mpd_public_amd import synthetic as syn
    from mpd_public_amd.parallel import expand_contexts
    st = torch.from_numpy(syn.synth_tensor("mc_s", (n_ctx, D), "uniform", 0.6)).cuda()
    gl = torch.from_numpy(syn.synth_tensor("mc_g", (n_ctx, D), "uniform", 0.6)).cuda()
    hs, hg = expand_contexts(st, gl, n)
    B = n_ctx * n

    def one():
        return dm.plan({0: hs, 63: hg}, B, 64, n0, None, lambda t: 0.5, return_chain=True, n_per_context=n)
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(plans):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / plans
    return {"workload": f"{n_ctx} contexts x {n} trajectories in ONE plan (cfg2 shape, unguided)", "plan_ms": round(dt * 1e3, 2),
            "ms_per_context": round(dt * 1e3 / n_ctx, 2), "context_denoising_steps_per_s": round((T + n0) * n_ctx / dt, 1)}


def planner_baseline_leg(n=100, opt_iters=500):
    """the reference's baseline planner pipeline (generate_trajectories.py: RRT-Connect initialisation + GPMP-objective
    optimisation) for ONE context of n trajectories, on this GPU, next to the diffusion sampler's plan time for the same n."""
    import torch
    from mpd_public_amd.generate_trajectories import generate_collision_free_trajectories as gen
    out = {}
    for env_id, robot in (("EnvNarrowPassageDense2D", "RobotPointMass"), ("EnvSpheres3D", "RobotPanda")):
        gen(env_id, robot, n, None, gpmp_opt_iters=opt_iters, seed=1)   # warm-up (allocator, kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_coll, n_free = gen(env_id, robot, n, None, gpmp_opt_iters=opt_iters, seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        last = gen.last
        out[f"{env_id}-{robot}"] = {"n_trajectories": n, "wall_ms": round(dt * 1e3, 1), "rrt_connect_ms": round(last["times"]["rrt_connect_s"] * 1e3, 1),
                                    "gpmp_ms": round(last["times"]["gpmp_s"] * 1e3, 1), "gpmp_iters": opt_iters, "rrt_solved": last["rrt_solved"],
                                    "fraction_free": round(last["fraction_free"], 3), "collision_intensity": round(last["collision_intensity"], 4)}
    return out


def training_leg(steps=100, B=32, T=25, D=4, opt=1, baseline=True):
    """training iterations per second at the reference's training configuration (train.py: batch 32, T = 25, dim_mults
    option 1, Adam 1e-4, clip_grad_norm 1.0, EMA every 10 steps) - the native step (HIP forward + backward + Adam + EMA) next to
    the same iteration written with torch autograd over the functional U-Net (ATen / MIOpen kernels) on the same GPU, which is
    how the reference trains."""
    import copy
    import torch
    import mpd_public_amd as m
    from mpd_public_amd import synthetic as syn
    from mpd_public_amd.trainer import TrainStep, EMA
    net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[opt])
    sd = syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})   # as build_model: shapes from the product
    net.load_state_dict(sd, strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True).cuda()
    ema_model = copy.deepcopy(dm)
    x0 = torch.from_numpy(syn.synth_tensor("train_x0", (B, 64, D), "uniform", 0.8)).cuda()
    hc = {0: x0[:, 0, :].contiguous(), 63: x0[:, -1, :].contiguous()}
    ts, ema = TrainStep(dm), EMA(0.995)

    def native(k):   # what trainer.train() runs per step with the native optimiser: TrainStep.step (one hipGraph replay from the third call on)
        ts.step(x0, hc, 1e-4, max_norm=1.0)
        if k % 10 == 0:
            ema.update_model_average(ema_model, dm)
    for k in range(8):   # (covers step()'s own measurement of the two launch forms: 6 calls)
        native(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        native(k)
    torch.cuda.synchronize()
    dt_native = (time.perf_counter() - t0) / steps
    rec = {"workload": f"p_losses + backward + clip_grad_norm_ + Adam + EMA/10, batch {B} x H=64 x D={D}, T={T}, dim_mults option {opt}, fp32",
           "launch_mode": {"note": "TrainStep.step measures both forms on this host + GPU (eager launches: calls 2-3; one hipGraph replay per iteration: "
                                   "calls 5-6) and keeps the faster (same launches, same device-side RNG / step count / lr: same numbers); MPDX_TRAIN_GRAPH=1 / 0 forces either",
                           "decided": ts.launch_mode()},
           "train_steps_per_s": round(1.0 / dt_native, 1), "ms_per_train_step": round(dt_native * 1e3, 3)}
    # roofline of the iteration: algorithmic FLOPs = 3 x the forward pass (forward + input gradients + weight gradients of every
    # convolution; GroupNorm / Mish / Adam are O(activations + parameters)), the forward count from the library's own layer table
    try:
        from mpd_public_amd import _lib
        lib = _lib.load()
        hdl, packed, tab, wsb = dm.model.engine(T, B)
        cap = 128
        ms_ = (C.c_float * cap)(); fl_ = (C.c_double * cap)(); nm_ = (C.c_char_p * cap)(); n_ = C.c_int()
        _lib.check(lib.mpdx_unet_profile(hdl, packed.data_ptr(), tab.data_ptr(), dm.model._timetab_T, x0.data_ptr(), 0, B, wsb.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream, cap, ms_, fl_, nm_, C.byref(n_)), "mpdx_unet_profile")
        fwd = float(sum(fl_[k] for k in range(n_.value)))
        tf = 3.0 * fwd / dt_native / 1e12
        rec["roofline"] = {"bound": "latency of a chain of ~50 dependent launches (4 whole-trajectory programs of 27-59 us, ~30 inner-level launches of 5-10 us; fp32 MFMA for the convolutions)",
                           "algorithmic_flop_per_iteration": 3.0 * fwd, "forward_flop": fwd, "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 4),
                           "launches_per_iteration": _train_launches_from_profile(B)}
    except Exception as e:   # the record must survive a failing helper
        rec["roofline"] = {"error": f"{type(e).__name__}: {e}"}
    if not baseline:
        return rec
    # the reference's way: autograd over ATen kernels + torch.optim.Adam, same GPU (the functional U-Net of oracle/ on CUDA tensors;
    # only this comparison leg touches oracle/)
    from oracle import unet as ounet
    from oracle import diffusion as odiff, schedules as osched
    params = {k: v.clone().cuda().requires_grad_(True) for k, v in sd.items()}
    opt_t = torch.optim.Adam(list(params.values()), lr=1e-4)
    buf = {k: v.cuda() for k, v in osched.make_buffers(T, "exponential").items()}
    ema_t = {k: v.detach().clone() for k, v in params.items()}

    def eager(k):
        t = torch.randint(0, T, (B,), device="cuda")
        noise = torch.randn_like(x0)
        x_noisy = odiff.apply_hard_conditioning(odiff.q_sample(buf, x0, t, noise), hc)
        x_recon = odiff.apply_hard_conditioning(ounet.unet_forward(params, x_noisy, t), hc)
        loss = torch.nn.functional.mse_loss(x_recon, noise)
        opt_t.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt_t.step()
        if k % 10 == 0:
            with torch.no_grad():
                for kk in ema_t:
                    ema_t[kk].mul_(0.995).add_(params[kk].detach(), alpha=0.005)
    for k in range(5):
        eager(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        eager(k)
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    rec["torch_autograd_same_gpu"] = {"train_steps_per_s": round(1.0 / dt_eager, 1), "ms_per_train_step": round(dt_eager * 1e3, 3),
                                      "what": "the same iteration as torch autograd over ATen/MIOpen kernels + torch.optim.Adam (how the reference trains)"}
    rec["speedup"] = round(dt_eager / dt_native, 2)
    return rec


# ------------------------------------------------------------------------------------------------------ the line the driver keeps
def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(out):
    """The LAST stdout line: the contract's keys + `roofline` + `cpu_baseline` + every number DESIGN.md section 3 quotes, FLAT, in < 6 KB (the
    full record is ~60 KB and the driver keeps an 8.7-KB tail - cfg3 / cfg4 / cfg5 and the training numbers fell off it).  The full record goes to
    bench_full.json beside this script and, as one line, to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "plan_wall_clock_ms", "per_rank_ms_per_step", "backend", "speedup_vs_cpu_baseline", "cpu_baseline_reason")
    c = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    c["config"] = {"workload": "cfg2: EnvDense2D-RobotPointMass shape, 100 trajectories x H=64 x D=4, T=100 (+5), unguided, dim_mults (1,2,4,8); one plan = one bench step"
                   if str(cfg.get("workload", "")).startswith("cfg2") else cfg.get("workload"),
                   "parallelism": cfg.get("parallelism"), "denoising_steps_per_plan": cfg.get("denoising_steps_per_plan"),
                   "trajectory_steps_per_s": cfg.get("trajectory_steps_per_s")}
    r = out.get("roofline")
    if isinstance(r, dict):
        c["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "avg_launch_us",
                                               "launches_per_unet_pass", "share_of_pass", "algorithmic_flop_per_launch", "frac_in_situ", "frac_bracketed", "frac_rocprof",
                                               "unet_pass_us", "unet_pass_tflops")}
        c["roofline"]["timed"] = ("in situ HIP events on the launch stream of this run: frac = differential (pass - pass without the class), frac_bracketed = "
                                  "event pair around the launches; frac_rocprof = committed rocprofv3 summary of these sources")
        c["roofline"]["rocprof_avg_launch_us"] = _get(r, "rocprof", "avg_launch_us")
        c["roofline"]["rocprof_file"] = _get(r, "rocprof", "file")
        c["roofline"]["traffic_source"] = r.get("traffic_source")
        c["roofline"]["frac_of_peak_at_occupancy"] = _get(r, "occupancy", "frac_of_peak_at_occupancy")
        c["roofline"]["cus_in_use"] = _get(r, "occupancy", "cus_in_use")
    pr = out.get("plan_roofline")
    if isinstance(pr, dict):
        c["plan_roofline"] = {k: pr.get(k) for k in ("hbm_GBps", "hbm_frac", "fp32_TFLOPs", "fp32_peak_frac", "algorithmic_bytes_per_step")}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "host_cores", "host_cpu", "plan_wall_s")}
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:200]
        c["cpu_baseline"]["multi_process_value"] = _get(cb, "multi_process", "value")
    else:
        c["cpu_baseline"] = cb
    # ---- flat figures of the sub-records
    g = out.get("guided") or {}
    for cfg_name in ("cfg3", "cfg4"):
        rec = g.get(cfg_name) or {}
        c[f"{cfg_name}_plan_ms"] = rec.get("plan_ms")
        c[f"{cfg_name}_steps_per_s"] = rec.get("denoising_steps_per_s")
        oc = rec.get("oracle_check") or {}
        e3 = oc.get("equal_to_3sf") or {}
        c[f"{cfg_name}_equal_to_3sf"] = [e3.get(k) for k in ("collision_free_rate", "collision_intensity", "path_length", "smoothness")] if e3 else oc.get("error")
        c[f"{cfg_name}_within_fp32_class"] = _get(oc, "chain_class", "within_fp32_class")
        c[f"{cfg_name}_flips_hip_vs_oracle32"] = [_get(oc, "chain_class", "flips_vs_fp64_chain", "hip"), _get(oc, "chain_class", "flips_vs_fp64_chain", "oracle_fp32")]
        c[f"{cfg_name}_ambiguous_waypoints"] = oc.get("ambiguous_waypoints")
    if "error" in g:
        c["guided_error"] = g["error"]
    c["guide_us_per_launch"] = {"2d": _get(g, "cfg3", "guide_kernel", "us_per_launch"), "panda": _get(g, "cfg4", "guide_kernel", "us_per_launch")}
    sh = out.get("sharded") or {}
    c["cfg5_shard_plan_ms"] = sh.get("plan_ms_per_rank_max")
    c["cfg5_steps_per_s"] = sh.get("denoising_steps_per_s")
    c["cfg5_fp32_peak_frac"] = sh.get("fp32_peak_frac")
    c["cfg5_fp32_TFLOPs"] = sh.get("fp32_TFLOPs")
    if "error" in sh:
        c["sharded_error"] = sh["error"]
    if sh.get("ranks", 1) and out.get("n_gpus", 1) > 1:
        c["rccl_world_size"] = sh.get("process_group_world_size")
        c["rccl_backend"] = sh.get("backend")
        c["rccl_version"] = sh.get("rccl_version")
        c["gather_ms"] = {"collective": _get(sh, "gather_ms", "all_gather_into_tensor", "max"),
                          "one_hop": _get(sh, "gather_ms", "one_hop_send_recv", "max", default=_get(sh, "gather_ms", "one_hop_send_recv", "error"))}
        c["checksums_ok"] = bool(sh.get("gather_verified")) if "error" not in sh else False
        c["cfg5_plan_ms_per_rank"] = _get(sh, "plan_ms_per_rank", "all")
        c["gather_bytes_per_rank"] = sh.get("all_gather_bytes_per_rank")
    sv = out.get("serving") or {}
    c["serving_ms_per_context"] = sv.get("ms_per_context")
    c["serving_plan_ms"] = sv.get("plan_ms")
    tr = out.get("training") or {}
    c["train_ms"] = {"b32_D4": tr.get("ms_per_train_step"), "b128_D14": _get(tr, "batch128_D14", "ms_per_train_step"),
                     "b512_D14": _get(tr, "batch512_D14", "ms_per_train_step"), "three_level_b128_D14": _get(tr, "three_level_batch128_D14", "ms_per_train_step")}
    c["train_fp32_peak_frac"] = {"b32_D4": _get(tr, "roofline", "frac"), "b128_D14": _get(tr, "batch128_D14", "fp32_peak_frac"),
                                 "b512_D14": _get(tr, "batch512_D14", "fp32_peak_frac")}
    c["train_launch_mode"] = [m_.get("mode") for m_ in (_get(tr, "launch_mode", "decided") or [])]
    c["train_launches_per_iteration"] = _get(tr, "roofline", "launches_per_iteration", "launches_per_iteration")
    c["train_launches_source"] = _get(tr, "roofline", "launches_per_iteration", "file")
    c["train_speedup_vs_torch_autograd"] = tr.get("speedup")
    if "error" in tr:
        c["training_error"] = tr["error"]
    td = out.get("trained") or {}
    c["trained"] = {"train_steps": td.get("train_steps"), "train_s": td.get("train_s"), "generate_s": td.get("generate_s"),
                    "loss_first_last": td.get("diffusion_loss_first_last"),
                    "free_rate_unguided": [p_.get("collision_free_rate") for p_ in _get(td, "plans", "diffusion_prior", default=[])],
                    "free_rate_guided": [p_.get("collision_free_rate") for p_ in _get(td, "plans", "mpd", default=[])],
                    "equal_to_3sf": (lambda e3: [e3.get(k) for k in ("collision_free_rate", "collision_intensity", "path_length", "smoothness")] if e3 else None)(
                        _get(td, "oracle_check", "equal_to_3sf")),
                    "within_fp32_class": _get(td, "oracle_check", "chain_class", "within_fp32_class"),
                    "check_free_rate_hip_o32_o64": [_get(td, "oracle_check", "plan_figures", k, "collision_free_rate") for k in ("hip", "oracle_fp32", "oracle_fp64")],
                    "error": td.get("error") or _get(td, "oracle_check", "error")}
    pb = out.get("planner_baseline") or {}
    c["planner_baseline_ms"] = {k.split("-")[0]: v.get("wall_ms") for k, v in pb.items() if isinstance(v, dict)}
    c["planner_baseline_free"] = {k.split("-")[0]: v.get("fraction_free") for k, v in pb.items() if isinstance(v, dict)}
    c["leg_seconds"] = out.get("leg_seconds")
    c["full_record"] = "bench_full.json beside bench.py; also one line on stderr"
    return c


def emit(out):
    """full record -> bench_full.json + stderr; compact line -> stdout (the last line, the one the driver parses and keeps)"""
    full = json.dumps(out)
    try:
        (ROOT / "bench_full.json").write_text(full + "\n")
        side = ROOT / "gpurun_out"
        if side.is_dir():   # (the GPU box merges gpurun_out/ back: the full record survives the box)
            (side / "bench_full.json").write_text(full + "\n")
    except OSError:
        pass
    print("# bench_full " + full, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(out))
    print(line, flush=True)



def _respawn(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `sharded` / `serving` sub-records")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        _respawn(args)
    t_main = time.perf_counter()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    rig = world > ndev   # development rig: more ranks than GPUs (e.g. --gpus 2 on a 1-GPU box) - ranks share GPUs, gloo rendezvous
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL over xGMI
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rig:   # RCCL refuses two ranks on one device; the rig exercises the sharding / gather / checksum code, not the fabric
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("NCCL_DEBUG", "VERSION")   # RCCL prints its version line at init (the driver's log keeps stderr)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))  # nccl == RCCL on ROCm
        if dist.get_world_size() != world or (args.gpus > 1 and dist.get_world_size() != args.gpus):
            raise SystemExit(f"process group of {dist.get_world_size()} ranks, WORLD_SIZE={world}, --gpus {args.gpus}")

    env_id, robot, D, mults, T, B, n0, guided, n_ctx = CONFIGS[args.config]
    dm, sd = build_model(D, mults, T, device)
    dm.manual_seed(30 + rank)
    from mpd_public_amd import synthetic as syn
    extra = lambda t: 0.5  # noqa: E731  inference.py:243
    guide_kw = build_guide(env_id, robot, T, device) if guided else {}
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
    per_rank_dt = [r[0] for r in _all_ranks(dist, [dt], device)]
    dt = max(per_rank_dt)
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
    if world > 1:
        out["per_rank_ms_per_step"] = [round(v / args.steps * 1e3, 3) for v in per_rank_dt]
        out["backend"] = dist.get_backend() + (" (single-GPU rig: ranks share a GPU, no xGMI traffic)" if rig else " (RCCL over xGMI)")
    leg_s = {"headline": round(time.perf_counter() - t_main, 1)}   # wall seconds of every leg (the driver's budget: where this script's minutes go)

    def timed(name, fn, *a, **k):
        t_ = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            leg_s[name] = round(time.perf_counter() - t_, 1)
    out["leg_seconds"] = leg_s
    if rank == 0 and not args.no_roofline:
        roof, unet_flops = timed("roofline", roofline_leg, dm, B, T)
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
            for c in roof["classes"]:
                print(f"# {c['us_per_pass']:9.1f} us {c['share']*100:5.1f}%  {c['launches_per_pass']:3d} launches  {c['tflops']:7.2f} TF/s "
                      f"({c['frac']*100:4.1f}% of peak)  {c['class']}", file=sys.stderr)
    if not args.no_extras and args.config == "cfg2":
        # every rank takes part in the sharded sub-record (it ends in a collective when N > 1).  At N > 1 a watchdog stands behind it:
        # should a collective of this leg never return (first contact with the 8-GPU fabric happens in the driver's run), rank 0 still
        # prints the headline line - already complete above - and every rank exits instead of blocking the driver.
        dog = _Watchdog(900.0, rank, out) if world > 1 else None
        try:
            rec = timed("sharded", sharded_leg, rank, world, dist, device)
            if rank == 0:
                out["sharded"] = rec
        except Exception as e:  # the headline line must survive a failing sub-record
            if rank == 0:
                out["sharded"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if dog is not None:
                dog.cancel()
        if rank == 0 and world == 1:
            try:
                out["guided"] = timed("guided", guided_leg, device)
            except Exception as e:
                out["guided"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["serving"] = timed("serving", serving_leg, dm, D, T, n0)
            except Exception as e:
                out["serving"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["planner_baseline"] = timed("planner_baseline", planner_baseline_leg)
            except Exception as e:
                out["planner_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["trained"] = timed("trained", trained_leg, device)
            except Exception as e:
                out["trained"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["training"] = timed("training_b32", training_leg)
                # the larger shapes of the same iteration (no torch-autograd leg): where the step stops being launch-bound
                for nm, (tb, td, tsteps) in {"batch128_D14": (128, 14, 100), "batch512_D14": (512, 14, 60)}.items():
                    r = timed("training_" + nm, training_leg, steps=tsteps, B=tb, D=td, baseline=False)
                    out["training"][nm] = {"ms_per_train_step": r["ms_per_train_step"], "train_steps_per_s": r["train_steps_per_s"],
                                           "fp32_TFLOPs": r.get("roofline", {}).get("achieved"), "fp32_peak_frac": r.get("roofline", {}).get("frac"),
                                           "launch_mode": r.get("launch_mode")}
                # the OTHER network the reference's launch script trains (launch_train_01.py:81-84: unet_dim_mults_option 0 = (1, 2, 4), batch 128)
                r = timed("training_3level_b128", training_leg, steps=100, B=128, D=14, opt=0, baseline=False)
                out["training"]["three_level_batch128_D14"] = {"ms_per_train_step": r["ms_per_train_step"], "train_steps_per_s": r["train_steps_per_s"],
                                                               "fp32_TFLOPs": r.get("roofline", {}).get("achieved"),
                                                               "fp32_peak_frac": r.get("roofline", {}).get("frac"), "launch_mode": r.get("launch_mode")}
            except Exception as e:
                out["training"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = timed("cpu_baseline", cpu_baseline_leg, sd, D, T, B, n0)
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    elif rank == 0:   # the key is always present: a record consumer must not fail on its shape at N > 1
        out["cpu_baseline"] = None
        out["cpu_baseline_reason"] = ("--no-cpu-baseline" if args.no_cpu_baseline else
                                      "measured on rank 0 at N=1 only (the contract: one bounded CPU sample per box, not one per rank)")
        for k in ("guided", "serving", "planner_baseline", "trained", "training"):
            out.setdefault(k, None)
        if world > 1:
            out["sub_records_reason"] = "guided / serving / planner_baseline / trained / training are single-GPU sub-records: N=1 only"
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
