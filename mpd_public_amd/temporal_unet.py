"""TemporalUnet - drop-in for mpd.models.diffusion_models.temporal_unet.TemporalUnet (temporal_unet.py:20-171).

Same constructor arguments, same module tree / state-dict keys (so a reference checkpoint loads with strict=True,
inference.py:145-148), same call protocol ``model(x[B,H,D], time[B] int64, context=None) -> eps[B,H,D]``.
The forward does NOT run the torch modules: they only hold the parameters.  On first use the parameters are
repacked once into MFMA fragment order and every call runs the hand-written gfx950 kernels in libmpdx.so.
Only the configuration the reference ever builds is supported: conditioning_type=None, self_attention=False
(inference.py:132-141, train.py:94-107).
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib

UNET_DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}  # temporal_unet.py:14-17


def group_norm_n_groups(n_channels, target_n_groups=8):  # layers.py:389-395
    if n_channels < target_n_groups:
        return 1
    for n_groups in range(target_n_groups, target_n_groups + 10):
        if n_channels % n_groups == 0:
            return n_groups
    return 1


class _Holder(nn.Module):
    """Parameter container; never executed (the kernels read the repacked weights)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder - TemporalUnet.forward runs libmpdx kernels, not torch modules")


def _seq_with(**at):
    """nn.Sequential whose integer slots hold the given modules, nn.Identity elsewhere (keeps state-dict indices)."""
    n = max(int(k) for k in at) + 1
    return nn.Sequential(*[at.get(str(i), nn.Identity()) for i in range(n)])


def _conv_block(cin, cout, k=5):
    m = _Holder()
    m.block = _seq_with(**{"0": nn.Conv1d(cin, cout, k, padding=k // 2), "2": nn.GroupNorm(group_norm_n_groups(cout), cout), "4": nn.Identity()})
    return m


def _res_block(cin, cout, cond_dim):
    m = _Holder()
    m.blocks = nn.ModuleList([_conv_block(cin, cout), _conv_block(cout, cout)])
    m.cond_mlp = _seq_with(**{"1": nn.Linear(cond_dim, cout), "2": nn.Identity()})
    m.residual_conv = nn.Conv1d(cin, cout, 1) if cin != cout else nn.Identity()
    return m


def _resample(dim, up):
    m = _Holder()
    m.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1) if up else nn.Conv1d(dim, dim, 3, 2, 1)
    return m


class TemporalUnet(nn.Module):
    def __init__(self, n_support_points=None, state_dim=None, unet_input_dim=32, dim_mults=(1, 2, 4, 8), time_emb_dim=32,
                 self_attention=False, conditioning_embed_dim=4, conditioning_type=None, attention_num_heads=2,
                 attention_dim_head=32, **kwargs):
        super().__init__()
        if conditioning_type not in (None, "None"):
            raise NotImplementedError("mpd_public_amd.TemporalUnet: only conditioning_type=None is supported "
                                      "(the only configuration the reference scripts build)")
        if self_attention:
            raise NotImplementedError("mpd_public_amd.TemporalUnet: self_attention=True is not supported")
        self.state_dim = state_dim
        self.conditioning_type = None
        self.n_support_points = n_support_points
        self.unet_input_dim = unet_input_dim
        self.dim_mults = tuple(int(m) for m in dim_mults)
        self.time_emb_dim = time_emb_dim

        dims = [state_dim] + [unet_input_dim * m for m in self.dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        n_res = len(in_out)

        self.time_mlp = _Holder()
        self.time_mlp.encoder = _seq_with(**{"1": nn.Linear(32, 32 * 4), "3": nn.Linear(32 * 4, time_emb_dim)})
        self.downs = nn.ModuleList()
        for i, (ci, co) in enumerate(in_out):
            last = i >= n_res - 1
            self.downs.append(nn.ModuleList([_res_block(ci, co, time_emb_dim), _res_block(co, co, time_emb_dim), nn.Identity(),
                                             nn.Identity(), _resample(co, up=False) if not last else nn.Identity()]))
        mid = dims[-1]
        self.mid_block1 = _res_block(mid, mid, time_emb_dim)
        self.mid_attn = nn.Identity()
        self.mid_attention = nn.Identity()
        self.mid_block2 = _res_block(mid, mid, time_emb_dim)
        self.ups = nn.ModuleList()
        for ci, co in reversed(in_out[1:]):  # every up stage upsamples (temporal_unet.py:98-107)
            self.ups.append(nn.ModuleList([_res_block(co * 2, ci, time_emb_dim), _res_block(ci, ci, time_emb_dim), nn.Identity(),
                                           nn.Identity(), _resample(ci, up=True)]))
        self.final_conv = nn.Sequential(_conv_block(unet_input_dim, unet_input_dim), nn.Conv1d(unet_input_dim, state_dim, 1))

        # device-side state (created lazily on the parameters' device)
        self._h = None          # mpdx_unet*
        self._packed = None
        self._timetab = None
        self._timetab_T = 0
        self._ws = None
        self._ws_B = 0
        self._stamp = None
        # the native model is created HERE (host-side only: no GPU needed), so that a configuration libmpdx cannot run - a GroupNorm
        # group of 2 or 64 channels, a width that is not a multiple of 16, a horizon the strided convolutions do not map back onto
        # itself - is refused at construction with the layer named, not at the first forward.  On a host where the library itself is
        # unavailable the module can still be BUILT (to inspect or convert a state_dict): the check then happens at the first engine() call,
        # which raises - there is no fallback.
        try:
            self._handle()
        except _lib.LibraryUnavailable:
            pass

    # ------------------------------------------------------------------------------------------- engine plumbing
    _DEVICE_STATE = {"_h": None, "_packed": None, "_timetab": None, "_timetab_T": 0, "_ws": None, "_ws_B": 0, "_stamp": None}

    def __getstate__(self):
        """copy.deepcopy (the reference's EMA pattern, trainer.py:67-85) and pickling copy the PARAMETERS only: the native
        handle (a ctypes pointer owned by this instance), the repacked weights, the time table and the workspace are
        dropped and rebuilt lazily by the copy's first engine() call - never shared, never double-freed."""
        state = dict(self.__dict__)
        state.update(self._DEVICE_STATE)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.update(self._DEVICE_STATE)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mpdx_unet_destroy(self._h)
        except Exception:
            pass

    def _handle(self):
        if self._h is None:
            lib = _lib.load()
            mults = (C.c_int32 * _lib.MAX_LEVELS)(*self.dim_mults)
            cfg = _lib.UnetCfg(int(self.state_dim), int(self.n_support_points), int(self.unet_input_dim), len(self.dim_mults),
                               mults, int(self.time_emb_dim))
            h = C.c_void_p()
            _lib.check(lib.mpdx_unet_create(C.byref(cfg), C.byref(h)), "mpdx_unet_create")
            self._h = h
        return self._h

    def _param_stamp(self):
        """(address, version counter) of every parameter: load_state_dict, an optimiser step, .to() and in-place edits all change it.  The parameter LIST is
        cached - walking the module tree costs 0.15 ms, which the step-by-step protocol loop paid per denoising step (round 6: 65.7 -> 21 ms per plan together
        with the other host-side items, tools/stepwise_probe.py) - and rebuilt by _apply (.to / .cuda / .float), after load_state_dict (assign=True replaces the
        Parameter objects) and on every 16th call (a Parameter object swapped in by hand)."""
        d = self.__dict__
        n = d.get("_stamp_calls", 0) + 1
        d["_stamp_calls"] = n
        pl = d.get("_plist")
        if pl is None or n % 16 == 0:
            pl = d["_plist"] = list(self.parameters())
        return tuple((p.data_ptr(), p._version) for p in pl)

    def _apply(self, fn, *a, **k):
        self.__dict__["_plist"] = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.__dict__["_plist"] = None
        return r

    def engine(self, T: int, B: int):
        """(handle, packed, timetab, workspace) ready for a batch of B trajectories and timesteps < T.
        Repacks when a parameter changed (load_state_dict after the first call)."""
        lib, h = _lib.load(), self._handle()
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("mpd_public_amd.TemporalUnet runs on an AMD GPU only (move the model to 'cuda'); there is no CPU fallback")
        st = _lib.current_stream()
        # (inside GaussianDiffusionModel.p_sample_loop the weights are checked ONCE, before the first step: `_weights_frozen`)
        frozen = self.__dict__.get("_weights_frozen", False) and self._packed is not None and self._stamp is not None
        stamp = self._stamp if frozen else self._param_stamp()
        if self._packed is None or self._stamp != stamp or self._packed.device != dev:
            packed = torch.zeros(lib.mpdx_unet_packed_floats(h), dtype=torch.float32, device=dev)
            sd = self.state_dict()
            n = lib.mpdx_unet_num_params(h)
            if n != len(sd):
                raise RuntimeError(f"state dict has {len(sd)} tensors, libmpdx expects {n}")
            for name, t in sd.items():
                t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
                _lib.check(lib.mpdx_unet_pack_param(h, name.encode(), t.data_ptr(), t.numel(), packed.data_ptr(), st), f"pack {name}")
            self._packed, self._stamp, self._timetab, self._timetab_T = packed, stamp, None, 0
        if self._timetab is None or self._timetab_T < T:
            T_tab = max(int(T), 128)
            tab = torch.empty(lib.mpdx_unet_timetab_floats(h, T_tab), dtype=torch.float32, device=dev)
            half = 16  # SinusoidalPosEmb(32): layers.py:249-251, computed by the host exactly as the reference does
            freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device=dev, dtype=torch.float32)
            _lib.check(lib.mpdx_unet_build_timetab(h, self._packed.data_ptr(), freqs.data_ptr(), T_tab, tab.data_ptr(), st), "timetab")
            torch.cuda.current_stream().synchronize()  # freqs is a temporary
            self._timetab, self._timetab_T = tab, T_tab
        if self._ws is None or self._ws_B < B or self._ws.device != dev:
            self._ws = torch.empty(lib.mpdx_unet_workspace_floats(h, B), dtype=torch.float32, device=dev)
            self._ws_B = B
        return h, self._packed, self._timetab, self._ws

    # ------------------------------------------------------------------------------------------- model protocol
    def forward(self, x, time, context=None):
        """x: [B,H,D] fp32, time: [B] int64 -> eps [B,H,D].  Batch-constant `time` (every caller on the planning path) is one
        kernel sequence; mixed timesteps run as one sequence per distinct value."""
        if context is not None:
            raise NotImplementedError("context conditioning is not supported (context is always None on this path, inference.py:182)")
        b, h, d = x.shape
        if h != self.n_support_points or d != self.state_dim:
            raise ValueError(f"expected [B,{self.n_support_points},{self.state_dim}], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        from .diffusion_model import timestep_hint
        hint = timestep_hint(time)   # (make_timesteps' tensors carry their batch-constant value: no host sync)
        tl = [hint] * int(time.numel()) if hint is not None and time.numel() in (1, b) else time.reshape(-1).tolist()  # one host sync, as sample_functions.py:28-29 has
        if len(tl) not in (1, b):
            raise ValueError(f"time must have 1 or {b} entries, got {len(tl)}")
        out = torch.empty_like(x)
        lib = _lib.load()

        def run(xg, t0, og):
            hdl, packed, tab, ws = self.engine(t0 + 1, xg.shape[0])
            _lib.check(lib.mpdx_unet_forward(hdl, packed.data_ptr(), tab.data_ptr(), self._timetab_T, xg.data_ptr(), t0,
                                             og.data_ptr(), xg.shape[0], ws.data_ptr(), _lib.current_stream()), "mpdx_unet_forward")

        distinct = sorted(set(int(v) for v in tl))
        if len(distinct) == 1:   # every caller on the planning path: one timestep for the whole batch
            run(x, distinct[0], out)
            return out
        # per-sample timesteps (the reference's training path): the time-conditioning tables are per integer t, and
        # trajectories are independent, so the batch is run in groups of equal t
        self.engine(distinct[-1] + 1, b)
        tt = torch.as_tensor(tl, device=x.device)
        for t0 in distinct:
            idx = (tt == t0).nonzero(as_tuple=True)[0]
            xg = x.index_select(0, idx).contiguous()
            og = torch.empty_like(xg)
            run(xg, t0, og)
            out.index_copy_(0, idx, og)
        return out
