"""Functional TemporalUnet forward (oracle; test infrastructure).

State-dict driven restatement of mpd/models/diffusion_models/temporal_unet.py:118-171 with the building
blocks of mpd/models/layers/layers.py (TimeEncoder :229-240, SinusoidalPosEmb :243-255, Downsample1d :258-264,
Upsample1d :267-273, Conv1dBlock :276-293, ResidualTemporalBlock :323-355, group_norm_n_groups :389-395) for the
only configuration any reference script builds: conditioning_type=None, self_attention=False.
"""
import math

import torch
import torch.nn.functional as F


def group_norm_n_groups(c: int, target: int = 8) -> int:
    # layers.py:389-395
    if c < target:
        return 1
    for g in range(target, target + 10):
        if c % g == 0:
            return g
    return 1


def sinusoidal_pos_emb(t: torch.Tensor, dim: int = 32) -> torch.Tensor:
    # layers.py:243-255 ; t may be int64 (the reference passes the long timestep tensor)
    half = dim // 2
    f = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, device=t.device) * -f)
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _cast(sd: dict, e: torch.Tensor) -> torch.Tensor:
    return e.to(next(iter(sd.values())).dtype)


def time_embedding(sd: dict, t: torch.Tensor) -> torch.Tensor:
    # layers.py:229-240: SinusoidalPosEmb(32) -> Linear(32,128) -> Mish -> Linear(128,32)
    e = _cast(sd, sinusoidal_pos_emb(t, 32))
    e = F.linear(e, sd["time_mlp.encoder.1.weight"], sd["time_mlp.encoder.1.bias"])
    e = F.mish(e)
    return F.linear(e, sd["time_mlp.encoder.3.weight"], sd["time_mlp.encoder.3.bias"])


def conv1d_block(sd: dict, p: str, x: torch.Tensor) -> torch.Tensor:
    # layers.py:283-290: Conv1d(k,pad=k//2) -> GroupNorm(n_groups, eps=1e-5, affine) -> Mish
    w = sd[p + ".block.0.weight"]
    y = F.conv1d(x, w, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    y = F.group_norm(y, group_norm_n_groups(w.shape[0]), sd[p + ".block.2.weight"], sd[p + ".block.2.bias"], eps=1e-5)
    return F.mish(y)


def residual_temporal_block(sd: dict, p: str, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
    # layers.py:343-355
    tb = F.linear(F.mish(temb), sd[p + ".cond_mlp.1.weight"], sd[p + ".cond_mlp.1.bias"])
    h = conv1d_block(sd, p + ".blocks.0", x) + tb[:, :, None]
    h = conv1d_block(sd, p + ".blocks.1", h)
    if (p + ".residual_conv.weight") in sd:
        res = F.conv1d(x, sd[p + ".residual_conv.weight"], sd[p + ".residual_conv.bias"])
    else:
        res = x
    return h + res


def n_resolutions(sd: dict) -> int:
    n = 0
    while f"downs.{n}.0.blocks.0.block.0.weight" in sd:
        n += 1
    return n


def unet_forward(sd: dict, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """x [B,H,D] fp32, t [B] int64 -> eps [B,H,D].  temporal_unet.py:118-171 (context=None)."""
    nres = n_resolutions(sd)
    temb = time_embedding(sd, t)
    h = x.transpose(1, 2)  # 'b h c -> b c h'  :138
    skips = []
    for i in range(nres):  # :141-150
        h = residual_temporal_block(sd, f"downs.{i}.0", h, temb)
        h = residual_temporal_block(sd, f"downs.{i}.1", h, temb)
        skips.append(h)
        if i < nres - 1:  # Downsample1d, Identity on the last level
            h = F.conv1d(h, sd[f"downs.{i}.4.conv.weight"], sd[f"downs.{i}.4.conv.bias"], stride=2, padding=1)
    h = residual_temporal_block(sd, "mid_block1", h, temb)  # :152-156
    h = residual_temporal_block(sd, "mid_block2", h, temb)
    for j in range(nres - 1):  # :158-165; every up stage upsamples, the level-0 skip is never popped
        h = torch.cat((h, skips.pop()), dim=1)
        h = residual_temporal_block(sd, f"ups.{j}.0", h, temb)
        h = residual_temporal_block(sd, f"ups.{j}.1", h, temb)
        h = F.conv_transpose1d(h, sd[f"ups.{j}.4.conv.weight"], sd[f"ups.{j}.4.conv.bias"], stride=2, padding=1)
    h = conv1d_block(sd, "final_conv.0", h)  # :167
    h = F.conv1d(h, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return h.transpose(1, 2)  # :169


def unet_param_shapes(state_dim: int, unet_input_dim: int = 32, dim_mults=(1, 2, 4, 8), time_emb_dim: int = 32) -> dict:
    """name -> shape of every TemporalUnet parameter (the module tree of temporal_unet.py:60-116)."""
    dims = [state_dim] + [unet_input_dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    shapes = {
        "time_mlp.encoder.1.weight": (128, 32), "time_mlp.encoder.1.bias": (128,),
        "time_mlp.encoder.3.weight": (time_emb_dim, 128), "time_mlp.encoder.3.bias": (time_emb_dim,),
    }

    def cblock(p, ci, co):
        shapes[p + ".block.0.weight"] = (co, ci, 5)
        shapes[p + ".block.0.bias"] = (co,)
        shapes[p + ".block.2.weight"] = (co,)
        shapes[p + ".block.2.bias"] = (co,)

    def rtb(p, ci, co):
        cblock(p + ".blocks.0", ci, co)
        cblock(p + ".blocks.1", co, co)
        shapes[p + ".cond_mlp.1.weight"] = (co, time_emb_dim)
        shapes[p + ".cond_mlp.1.bias"] = (co,)
        if ci != co:
            shapes[p + ".residual_conv.weight"] = (co, ci, 1)
            shapes[p + ".residual_conv.bias"] = (co,)

    nres = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        rtb(f"downs.{i}.0", ci, co)
        rtb(f"downs.{i}.1", co, co)
        if i < nres - 1:
            shapes[f"downs.{i}.4.conv.weight"] = (co, co, 3)
            shapes[f"downs.{i}.4.conv.bias"] = (co,)
    mid = dims[-1]
    rtb("mid_block1", mid, mid)
    rtb("mid_block2", mid, mid)
    for j, (ci, co) in enumerate(reversed(in_out[1:])):
        rtb(f"ups.{j}.0", co * 2, ci)
        rtb(f"ups.{j}.1", ci, ci)
        shapes[f"ups.{j}.4.conv.weight"] = (ci, ci, 4)  # ConvTranspose1d: [C_in, C_out, k]
        shapes[f"ups.{j}.4.conv.bias"] = (ci,)
    cblock("final_conv.0", unet_input_dim, unet_input_dim)
    shapes["final_conv.1.weight"] = (state_dim, unet_input_dim, 1)
    shapes["final_conv.1.bias"] = (state_dim,)
    return shapes
