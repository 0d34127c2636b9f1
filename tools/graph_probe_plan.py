"""dev probe: a whole plan replayed as a hipGraph (torch.cuda.CUDAGraph capture of dm.plan), one chain vs two concurrent sub-batch chains
(MPDX_PLAN_CHAINS is read once per process: run it once per setting).  Timing only - the captured noise is frozen."""
import sys, time, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
env_id, robot, D, mults, T, B, n0, guided, n_ctx = bench.CONFIGS[cfg]
dm, sd = bench.build_model(D, mults, T, "cuda")
dm.manual_seed(30)
from mpd_public_amd import synthetic as syn
hc = {0: torch.from_numpy(syn.synth_tensor("bench_hc0", (D,), "uniform", 0.6)).cuda(), 63: torch.from_numpy(syn.synth_tensor("bench_hc1", (D,), "uniform", 0.6)).cuda()}
gk = bench.build_guide(env_id, robot, T, "cuda") if guided else {}
noise = torch.randn(T + n0 + 1, B, 64, D, device="cuda")
def plan():
    return dm.plan(hc, B, 64, n0, noise, lambda t: 0.5, return_chain=True, **gk)
for _ in range(3): plan()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): plan()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 10 * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): plan()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = plan()
torch.cuda.synchronize()
cap = (time.perf_counter() - t0) * 1e3
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 10 * 1e3
print(f"{cfg} chains_env={os.environ.get('MPDX_PLAN_CHAINS', 'default')}  eager {eager:.3f} ms  graph replay {graph:.3f} ms  (capture+instantiate {cap:.1f} ms)  finite={bool(torch.isfinite(out[0]).all())}")
