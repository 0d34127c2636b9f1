"""Host-side bookkeeping of the drop-in classes (no GPU): the cached weight / schedule stamps that replaced a module-tree walk per denoising step
(round 6: the step-by-step protocol loop was host-bound), and the loop index that rides on make_timesteps' tensors."""
import torch
import torch.nn as nn

import mpd_public_amd as m
from mpd_public_amd.diffusion_model import make_timesteps


def _net():
    return m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[0])


def test_weight_stamp_sees_every_way_the_weights_can_change():
    """TemporalUnet.engine() repacks when the stamp changes: in-place updates (an optimiser step, load_state_dict's copy_), moved storage (.to / .double),
    load_state_dict(assign=True) (new Parameter objects) at once; a Parameter object swapped in by hand within 16 calls (the cached list's refresh)."""
    net = _net()
    s0 = net._param_stamp()
    assert net._param_stamp() == s0   # nothing changed: same stamp, from the cached list
    with torch.no_grad():
        next(net.parameters()).add_(1.0)   # what optimizer.step() does
    s1 = net._param_stamp()
    assert s1 != s0
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd)   # copy_ into the same tensors: version counters move
    s2 = net._param_stamp()
    assert s2 != s1
    net.load_state_dict({k: v.clone() for k, v in sd.items()}, assign=True)   # the Parameter objects themselves are replaced
    s3 = net._param_stamp()
    assert s3 != s2 and {p for p, _ in s3}.isdisjoint({p for p, _ in s2})
    net.double()   # _apply: new storage
    s4 = net._param_stamp()
    assert s4 != s3
    net.float()
    s5 = net._param_stamp()
    lin = net.time_mlp.encoder[1]
    lin.weight = nn.Parameter(torch.zeros_like(lin.weight))   # by hand: nothing tells the cache
    seen = False
    for _ in range(16):
        seen = seen or net._param_stamp() != s5
    assert seen


def test_schedule_buffers_host_copy_follows_the_buffers():
    """GaussianDiffusionModel.host_buffers(): the CPU copies the step scalars are read from are rebuilt when a schedule buffer changes (load_state_dict of a
    checkpoint with another schedule), with the buffer LIST cached."""
    dm = m.GaussianDiffusionModel(model=_net(), n_diffusion_steps=25, predict_epsilon=True)
    h0 = dm.host_buffers()
    b0 = float(h0["betas"][3])
    assert dm.host_buffers() is h0
    other = m.GaussianDiffusionModel(model=_net(), n_diffusion_steps=25, predict_epsilon=True, variance_schedule="cosine")
    sd = dm.state_dict()
    for k, v in other.state_dict().items():
        if not k.startswith("model."):
            sd[k] = v.clone()
    dm.load_state_dict(sd)
    h1 = dm.host_buffers()
    assert h1 is not h0 and float(h1["betas"][3]) == float(other.betas[3]) != b0
    assert abs(float(h1["noise_scale"][5]) - float(torch.exp(0.5 * other.posterior_log_variance_clipped[5]))) < 1e-7


def test_make_timesteps_carries_the_loop_index():
    """diffusion_model_base.py:12-15 make_timesteps: the same tensor, plus the loop index as a Python attribute - ddpm_sample_fn / TemporalUnet.forward read it
    instead of `int(t[0])` (a host sync per denoising step in the reference, sample_functions.py:28)."""
    t = make_timesteps(5, 17, "cpu")
    assert t.dtype == torch.long and t.shape == (5,) and bool((t == 17).all()) and t._mpdx_value == 17
    assert make_timesteps(2, -3, "cpu")._mpdx_value == -3   # the n_diffusion_steps_without_noise tail
    from mpd_public_amd.diffusion_model import timestep_hint
    assert timestep_hint(t) == 17 and timestep_hint(torch.full((5,), 17)) is None   # any other tensor: read with a sync, as the reference does
    t -= 1   # written to since: the hint is stale and dropped (the tensor says 16 now)
    assert timestep_hint(t) is None and int(t[0]) == 16
