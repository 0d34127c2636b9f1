#!/usr/bin/env python3
"""Why does a Panda guide launch take ~208 us inside a cfg5 plan and ~170 us in bench.py's stand-alone timing?  
The stand-alone figure (mpdx_guide_time) runs the kernel in GRADIENT-ONLY mode on one unchanging input; the plan runs it in APPLY mode (x updated in
place, hard conditions, per-context max|x| atomics; the last iteration of a guided step also adds the step's noise and writes the chain row).  This
probe times, at B = 6400 (128 contexts x 50), with HIP events around runs of back-to-back launches:
   (a) gradient-only, one input               (mpdx_guide_time: the bench's figure)
   (b) apply mode, x evolving, own flags      (mpdx_guide_step, 5 launches per group as in a guided step)
   (c) apply mode with a U-Net-sized kernel between the groups (cold L2 for x / the primitive table: the plan's situation)
needs a GPU; prints one line per variant."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import mpd_public_amd as m
from mpd_public_amd import _lib
from helpers import product_guide, obstacle_hugging_trajs

B, npc = (int(sys.argv[1]) if len(sys.argv) > 1 else 6400), 50
ds = m.TrajectoryDataset("EnvSpheres3D", "RobotPanda", tensor_args={"device": "cuda", "dtype": torch.float32})
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
x0 = obstacle_hugging_trajs(ds, B, seed="trace", scale=0.95).cuda()
pg = product_guide(ds).cuda()
gp = pg.device_params(x0.device)
nctx = max(1, B // npc)
flag_in = torch.zeros(nctx, dtype=torch.int32, device="cuda")
_lib.check(lib.mpdx_absmax(x0.data_ptr(), flag_in.data_ptr(), npc, B, 64, ds.state_dim, st))
g = torch.zeros_like(x0)
ms = C.c_float(0)
best = 1e9
for _ in range(3):
    _lib.check(lib.mpdx_guide_time(C.byref(gp), x0.data_ptr(), g.data_ptr(), flag_in.data_ptr(), npc, B, 64, ds.state_dim, 50, st, C.byref(ms)))
    best = min(best, ms.value)
print(f"(a) gradient-only, one input, 50 back-to-back launches: {best * 1e3:7.1f} us per launch")
hs, hg = x0[:, 0, :].contiguous(), x0[:, -1, :].contiguous()


def apply_groups(groups, between=None, amax=True, hard=True, evolve=True):
    x = x0.clone()
    flags = torch.zeros((groups * 5 + 1, nctx), dtype=torch.int32, device="cuda")
    flags[0] = flag_in
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot, k = 0.0, 0
    for gidx in range(groups):
        if between is not None:
            between()
        if not evolve:
            x.copy_(x0)
        e0.record()
        for it in range(5):
            _lib.check(lib.mpdx_guide_step(C.byref(gp), x.data_ptr(), None, hs.data_ptr() if hard else None, hg.data_ptr() if hard else None,
                                           flags[k if evolve else 0].data_ptr(), flags[k + 1].data_ptr() if amax else None, npc, B, 64, ds.state_dim, st))
            k += 1
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / (groups * 5) * 1e3


apply_groups(2)
print(f"(b) apply mode, x evolving, groups of 5 launches:       {min(apply_groups(10) for _ in range(3)):7.1f} us per launch")
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")   # 1 GiB: sweeps the L2s and most of the MALL between the groups
print(f"(c) the same with a 1-GiB fill between the groups:       {min(apply_groups(10, lambda: big.fill_(1.0)) for _ in range(3)):7.1f} us per launch")
print(f"(d) apply mode without the max|x| atomics (amax_out = NULL):  {min(apply_groups(10, amax=False) for _ in range(3)):7.1f} us per launch")
print(f"(e) apply mode without hard conditions:                     {min(apply_groups(10, hard=False) for _ in range(3)):7.1f} us per launch")
print(f"(f) apply mode, x reset to the probe input before each group: {min(apply_groups(10, evolve=False) for _ in range(3)):7.1f} us per launch")
