#!/usr/bin/env python3
"""U-Net pass time, fused segments vs per-layer launches, over batch sizes (dev tool; run once per MPDX_FUSED value)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from bench import build_model
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
for B in (100, 200, 400, 800, 1600, 3200, 6400):
    x = torch.randn(B, 64, 4, device="cuda"); t = torch.full((B,), 50, device="cuda", dtype=torch.long)
    for _ in range(3): dm.model(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): dm.model(x, t)
    torch.cuda.synchronize()
    print(f"B={B:5d}  {1e6*(time.perf_counter()-t0)/n:9.1f} us per U-Net pass")
