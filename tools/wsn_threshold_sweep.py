#!/usr/bin/env python3
"""U-Net pass time (D = 14) over the batch size with the round-5 weight-stationary kernels switched on / off: where conv_wsn (128 -> 128, Upsample1d)
and conv_wsp (128 -> 256 + 1x1) start to pay against the per-layer kernels (their prologue loads a wave's whole weight slice: 40-48 KB per wave).
dev tool, needs a GPU:  python tools/wsn_threshold_sweep.py"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import mpd_public_amd as m
from helpers import synth_sd, DIM_MULTS

net = m.TemporalUnet(n_support_points=64, state_dim=14, unet_input_dim=32, dim_mults=DIM_MULTS[1])
net.load_state_dict(synth_sd(14, 1), strict=True)
net = net.cuda().eval()


def pass_us(B, reps=30):
    x = torch.randn(B, 64, 14, device="cuda")
    t = torch.full((B,), 7, dtype=torch.long, device="cuda")
    for _ in range(3):
        net(x, t, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        net(x, t, None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("B      all-on   wsn-off  wsp-off  both-off   (us per U-Net pass)")
for B in (512, 640, 768, 1024, 1280, 1536, 2048, 3200, 6400):
    row = []
    for wsn, wsp in ((1, 1), (0, 1), (1, 0), (0, 0)):
        os.environ["MPDX_WSN"], os.environ["MPDX_WSP"] = str(wsn), str(wsp)
        row.append(min(pass_us(B) for _ in range(3)))
    print(f"{B:5d}  {row[0]:8.1f} {row[1]:8.1f} {row[2]:8.1f} {row[3]:8.1f}   wsn gain {row[1]-row[0]:+7.1f}  wsp gain {row[2]-row[0]:+7.1f}", flush=True)
