"""dev probe: unguided plan (T = 100 + 0, fused chain) on the THREE-level network (dim_mults option 0), ms per plan at a few batch sizes.
usage: python tools/plan_opt0_probe.py [B ...]      (MPDX_NO_MID3=1: downs.2 + the middle blocks one launch per layer, as before round 6)"""
import sys, time, torch
sys.path[:0] = ['.', 'tests']
from bench import build_model
for B in [int(v) for v in sys.argv[1:]] or [100, 800, 6400]:
    dm, sd = build_model(4, (1, 2, 4), 100, "cuda")
    hc = {0: torch.zeros(4, device="cuda"), 63: torch.ones(4, device="cuda") * 0.5}
    def plan():
        return dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=False)
    for _ in range(3): plan()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10 if B <= 800 else 3
    for _ in range(n): plan()
    torch.cuda.synchronize(); print(B, round((time.perf_counter() - t0) / n * 1e3, 3), "ms per plan (100 steps)")
