"""Dev probe: the repack launch of a training iteration (pack_train_kernel) on its own - both packs, and the forward pack only (packedT = null)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpd_public_amd as m
from mpd_public_amd import synthetic as syn, _lib
from mpd_public_amd.trainer import TrainStep

opt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[opt])
sd = syn.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict(sd, strict=True)
dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True).cuda()
ts = TrainStep(dm)
lib = _lib.load()
h = ts.unet._handle()
packed = ts._packed()
print("floats: flat", ts.fp.flat.numel(), "packed", packed.numel(), "packedT", ts.fp.packedT.numel())
def run(pt, n=200):
    st = _lib.current_stream()
    for _ in range(10): lib.mpdx_train_pack(h, ts.fp.flat.data_ptr(), packed.data_ptr(), pt, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.mpdx_train_pack(h, ts.fp.flat.data_ptr(), packed.data_ptr(), pt, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("both packs   %.2f us per launch (back to back)" % run(ts.fp.packedT.data_ptr()))
print("forward only %.2f us" % run(None))
