#!/usr/bin/env python3
"""Plan wall-clock vs number of start/goal contexts batched into ONE plan (cfg2 shape: 100 trajectories per context,
T=100+5, unguided).  Serving note for DESIGN.md (dev tool, needs a GPU)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
from mpd_public_amd import synthetic as syn
from mpd_public_amd.parallel import expand_contexts
D, T, n, n0 = 4, 100, 100, 5
dm, _ = build_model(D, (1, 2, 4, 8), T, "cuda")
for C_ in (1, 2, 4, 5, 8, 16):
    st = torch.from_numpy(syn.synth_tensor("mc_s", (C_, D), "uniform", 0.6)).cuda()
    gl = torch.from_numpy(syn.synth_tensor("mc_g", (C_, D), "uniform", 0.6)).cuda()
    hs, hg = expand_contexts(st, gl, n)
    B = C_ * n
    def one():
        return dm.plan({0: hs, 63: hg}, B, 64, n0, None, lambda t: 0.5, return_chain=True, n_per_context=n)
    for _ in range(2): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{C_:3d} contexts x {n} = {B:5d} trajectories: {dt*1e3:7.2f} ms per plan   {dt*1e3/C_:6.2f} ms per context   {(T+n0)*C_/dt:8.0f} context-steps/s")
