"""dev probe: the step-by-step PROTOCOL loop (run_inference(fused=False): eps-model call -> sample_fn per step, the reference's own loop shape) against the
fused plan, ms per plan and host enqueue ms; cfg 2 shape (B = 100, D = 4, T = 100 + 5) and the guided 2-D shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dm, sd = build_model(4, (1, 2, 4, 8), 100, "cuda")
hc = {0: torch.zeros(4, device="cuda"), 63: torch.ones(4, device="cuda") * 0.5}
for fused in (True, False):
    f = lambda: dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=False, fused=fused, n_diffusion_steps_without_noise=5)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fused={fused}: enqueue {(t1 - t0) / 5 * 1e3:.2f} ms per plan, complete {(t2 - t0) / 5 * 1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=False, fused=False, n_diffusion_steps_without_noise=5)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
