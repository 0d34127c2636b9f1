import time, torch, copy, sys
sys.path.insert(0,'.')
import mpd_public_amd as m
from mpd_public_amd import synthetic as syn
from mpd_public_amd.trainer import TrainStep, EMA
for (B,D) in ((32,4),(128,14)):
    net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[1])
    sd = syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd, strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True).cuda()
    x0 = torch.from_numpy(syn.synth_tensor("train_x0", (B, 64, D), "uniform", 0.8)).cuda()
    hc = {0: x0[:, 0, :].contiguous(), 63: x0[:, -1, :].contiguous()}
    ts = TrainStep(dm)
    for k in range(5): ts.loss_backward(x0, hc); ts.adam_step(1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for k in range(100): ts.loss_backward(x0, hc); ts.adam_step(1e-4, max_norm=1.0)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(B, D, 'enqueue %.3f ms/it, total %.3f ms/it'%((t1-t0)*10,(t2-t0)*10))
