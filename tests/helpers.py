"""Shared test helpers: synthetic weights, toy costs (same formulas as tests/golden/make_golden.py)."""
import numpy as np
import torch

from mpd_public_amd import synthetic as syn
from oracle.unet import unet_param_shapes

DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}


def synth_sd(D, opt):
    return syn.synth_state_dict(unet_param_shapes(D, 32, DIM_MULTS[opt]))


def toy_cost(x, x_interpolated=None, return_invidual_costs_and_weights=False, **kw):
    q = x.shape[-1] // 2
    c1 = (x_interpolated[..., :q] - 0.3).pow(2).sum((-1, -2)) * 3.0
    c2 = (x[:, 1:, :] - x[:, :-1, :]).pow(2).sum((-1, -2)) * 0.5 + (x[..., q:]).abs().sum((-1, -2)) * 0.01
    return [c1, c2], [1e-2, 3e-3]


def t(name, shape, kind="normal", scale=1.0):
    return torch.from_numpy(syn.synth_tensor(name, shape, kind, scale))


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
