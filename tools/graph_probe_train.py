"""dev probe: does replaying the training step as a hipGraph beat eager launches?  (timing only: the captured scalars are frozen)"""
import time, torch, sys
sys.path.insert(0, ".")
import mpd_public_amd as m
from mpd_public_amd import synthetic as syn
from mpd_public_amd.trainer import TrainStep
from oracle import unet as ounet
B, D, T, opt = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 25, 1
net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[opt])
net.load_state_dict(syn.synth_state_dict(ounet.unet_param_shapes(D, 32, m.UNET_DIM_MULTS[opt])), strict=True)
dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True).cuda()
x0 = torch.from_numpy(syn.synth_tensor("train_x0", (B, 64, D), "uniform", 0.8)).cuda()
hc = {0: x0[:, 0, :].contiguous(), 63: x0[:, -1, :].contiguous()}
ts = TrainStep(dm)
t = torch.randint(0, T, (B,), device="cuda")
noise = torch.randn_like(x0)
def step():
    ts.loss_backward(x0, hc, t=t, noise=noise)
    ts.adam_step(1e-4, max_norm=1.0)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("eager  ms/step", (time.perf_counter() - t0) / 50 * 1e3)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize()
print("graph  ms/step", (time.perf_counter() - t0) / 50 * 1e3)
