#!/usr/bin/env python3
"""Compare the fused-segment path against the per-layer path on the GPU (dev tool): run once per MPDX_FUSED_MASK."""
import os, sys, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    import torch
    from helpers import synth_sd, t, DIM_MULTS
    import mpd_public_amd as m
    D, opt, B = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    net = m.TemporalUnet(n_support_points=64, state_dim=D, dim_mults=DIM_MULTS[opt]); net.load_state_dict(synth_sd(D, opt)); net = net.cuda()
    x = t("bisect_x", (B, 64, D)).cuda()
    y = net(x, torch.full((B,), 7, device="cuda", dtype=torch.long))
    torch.save(y.cpu(), sys.argv[5])
    sys.exit(0)
import torch
for D, opt in ((4, 0), (4, 1), (14, 1)):
    outs = {}
    for mask in ("0", "1", "2", "4", "8", "15"):
        f = f"/tmp/bisect_{D}_{opt}_{mask}.pt"
        env = dict(os.environ, MPDX_FUSED_MASK=mask, MPDX_FUSED="1" if mask != "0" else "0")
        r = subprocess.run([sys.executable, __file__, "child", str(D), str(opt), "3", f], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(D, opt, mask, "FAILED", r.stderr[-400:]); continue
        outs[mask] = torch.load(f)
    for mask, y in outs.items():
        print(f"D={D} opt={opt} mask={mask:>2}: max|diff vs per-layer| = {(y - outs['0']).abs().max().item():.3e}")
