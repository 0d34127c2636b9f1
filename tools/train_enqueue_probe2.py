"""dev probe: is the batch-32 training iteration HOST-bound?  ms per iteration to ENQUEUE (no sync) vs to complete, graph mode, a few repeats."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpd_public_amd as m
from mpd_public_amd import synthetic as syn
from mpd_public_amd.trainer import TrainStep
B, D, opt = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 1
net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[opt])
net.load_state_dict(syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True).cuda()
x0 = torch.from_numpy(syn.synth_tensor("train_x0", (B, 64, D), "uniform", 0.8)).cuda()
hc = {0: x0[:, 0, :].contiguous(), 63: x0[:, -1, :].contiguous()}
ts = TrainStep(dm)
for k in range(8): ts.step(x0, hc, 1e-4, max_norm=1.0)
torch.cuda.synchronize()
for rep in range(5):
    t0 = time.perf_counter()
    for k in range(200): ts.step(x0, hc, 1e-4, max_norm=1.0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"batch {B}: enqueue {(t1 - t0) / 200 * 1e3:.3f} ms per iteration, complete {(t2 - t0) / 200 * 1e3:.3f} ms  ({ts.launch_mode()[0]['mode']})")
# split of the host time: the replay call alone, the input copies alone, step()'s Python around them
g = list(ts._graphs.values())[-1]
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(200): g["graph"].replay()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"graph.replay() alone: {(t1 - t0) / 200 * 1e3:.3f} ms per call (host)")
t0 = time.perf_counter()
for k in range(200): torch._foreach_copy_([g["x"], g["hc"][0], g["hc"][63]], [x0, hc[0], hc[63]])
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"_foreach_copy_ of the inputs alone: {(t1 - t0) / 200 * 1e3:.3f} ms per call (host)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(200): ts.step(x0, hc, 1e-4, max_norm=1.0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
