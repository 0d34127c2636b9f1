"""dev probe: the step-by-step PROTOCOL loop (run_inference(fused=False): eps-model call -> sample_fn -> guide(x) per step, the reference's own loop shape)
against the fused plan - ms per plan and host enqueue ms - at the cfg 2 shape (unguided) and the guided cfg 3 / cfg 4 shapes.
usage: python tools/stepwise_probe.py [profile]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mpd_public_amd import synthetic as syn
for cfg in ("cfg2", "cfg3", "cfg4"):
    env_id, robot, D, mults, T, B, n0, _, _ = bench.CONFIGS[cfg]
    dm, sd = bench.build_model(D, mults, T, "cuda")
    dm.manual_seed(30)
    gk = bench.build_guide(env_id, robot, T, "cuda") if cfg != "cfg2" else {}
    hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform", 0.6)).cuda(), 63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform", 0.6)).cuda()}
    for fused in (True, False):
        f = lambda: dm.run_inference(None, hc, n_samples=B, horizon=64, return_chain=False, fused=fused, n_diffusion_steps_without_noise=n0,
                                     noise_std_extra_schedule_fn=lambda t: 0.5, **gk)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{cfg} fused={fused}: enqueue {(t1 - t0) / 5 * 1e3:.2f} ms per plan, complete {(t2 - t0) / 5 * 1e3:.2f} ms", flush=True)
    if len(sys.argv) > 1 and cfg == "cfg4":
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(3): f()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
