"""dev probe: host time of the training forms that do NOT go through TrainStep.step's graph - (a) TrainStep.loss_backward + adam_step (eager launches),
(b) the reference's own loop shape: model.loss(...) -> loss.backward() -> clip_grad_norm_ -> torch.optim.Adam.step() (the autograd bridge; trainer.py:236-275)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpd_public_amd as m
from mpd_public_amd import synthetic as syn
from mpd_public_amd.trainer import TrainStep
B, D = 32, 4
net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[1])
net.load_state_dict(syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True).cuda()
x0 = torch.from_numpy(syn.synth_tensor("train_x0", (B, 64, D), "uniform", 0.8)).cuda()
hc = {0: x0[:, 0, :].contiguous(), 63: x0[:, -1, :].contiguous()}
ts = TrainStep(dm)
def a():
    ts.loss_backward(x0, hc); ts.adam_step(1e-4, max_norm=1.0)
opt = torch.optim.Adam(dm.parameters(), lr=1e-4)
def b():
    loss, info = dm.loss(x0, None, hc)
    opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(dm.parameters(), 1.0); opt.step()
for name, f in (("TrainStep.loss_backward + adam_step (eager)", a), ("model.loss -> backward -> clip -> torch Adam", b)):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: enqueue {(t1 - t0) / 100 * 1e3:.3f} ms per iteration, complete {(t2 - t0) / 100 * 1e3:.3f} ms", flush=True)
    if len(sys.argv) > 1:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50): f()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
