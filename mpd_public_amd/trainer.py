"""Training step on the GPU - host mirror of mpd/trainer/trainer.py (train :120-320, EMA :67-85, get_num_epochs :16-17,
save_models_to_disk :20-37) and mpd/losses/gaussian_diffusion_loss.py (GaussianDiffusionLoss :6-28) for the one model family the
reference trains: GaussianDiffusionModel over a TemporalUnet without context.

What the reference does per step with torch autograd + torch.optim.Adam,

    loss = model.loss(traj_normalized, context, hard_conds);  loss.backward();  clip_grad_norm_;  optimizer.step();  EMA

runs here as hand-written HIP kernels behind libmpdx.so (include/mpdx.h "training step"; kernels mpd_public_amd/csrc/train.hpp):
the U-Net forward with saved activations, its whole backward pass, the loss gradient, Adam and the EMA.  PyTorch provides the
device memory: parameters, gradients, Adam moments and the EMA copy are flat fp32 tensors, and the nn.Parameters of the
TemporalUnet are re-pointed at views of the flat parameter / gradient tensors (FlatParams) - `p.grad` is populated after
`loss_backward`, so `torch.optim` optimisers passed through `optimizers=` keep working exactly as in the reference.

There is no CPU fallback: every entry point raises if the model is not on an AMD GPU.
"""
from __future__ import annotations

import copy
import ctypes as C
import math
import weakref
import os
from math import ceil

import numpy as np
import torch

from . import _lib


def get_num_epochs(num_train_steps, batch_size, dataset_len):   # trainer.py:16-17
    return ceil(num_train_steps * batch_size / dataset_len)


class EarlyStopper:
    """trainer.py:45-64 (patience -1 deactivates it)."""

    def __init__(self, patience=10, min_delta=0):
        self.patience = patience
        self.min_delta = min_delta
        self.counter = 0
        self.min_validation_loss = float("inf")

    def early_stop(self, validation_loss):
        if self.patience == -1:
            return False
        if validation_loss < self.min_validation_loss:
            self.min_validation_loss = validation_loss
            self.counter = 0
        elif validation_loss > (self.min_validation_loss + self.min_delta):
            self.counter += 1
            if self.counter >= self.patience:
                return True
        return False


class FlatParams:
    """The parameters of a TemporalUnet as ONE flat fp32 tensor in the layout libmpdx differentiates (parameter i of
    mpdx_unet_param_info at mpdx_train_param_offset(i)), with every nn.Parameter (and its .grad) a view into it."""

    def __init__(self, unet):
        self.unet = unet
        lib, h = _lib.load(), unet._handle()
        dev = next(unet.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("training runs on an AMD GPU only (move the model to 'cuda'); there is no CPU fallback")
        self.n = int(lib.mpdx_train_flat_floats(h))
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.slices = {}
        self.order = []
        self.plist = []   # the Parameter objects in `order` (valid while aliased(): the autograd bridge walks this list instead of the module tree)
        named = dict(unet.named_parameters())
        name_p, shape, ndim = C.c_char_p(), (C.c_int32 * 3)(), C.c_int32()
        off, cnt = C.c_size_t(), C.c_size_t()
        for i in range(lib.mpdx_unet_num_params(h)):
            _lib.check(lib.mpdx_unet_param_info(h, i, C.byref(name_p), C.byref(shape), C.byref(ndim)), "param_info")
            _lib.check(lib.mpdx_train_param_offset(h, i, C.byref(off), C.byref(cnt)), "param_offset")
            name = name_p.value.decode()
            p = named[name]
            if p.numel() != cnt.value:
                raise RuntimeError(f"{name}: {p.numel()} elements, libmpdx expects {cnt.value}")
            view = self.flat[off.value:off.value + cnt.value].view(p.shape)
            view.copy_(p.data.to(torch.float32))
            p.data = view
            p.grad = self.grad[off.value:off.value + cnt.value].view(p.shape)
            self.slices[name] = (off.value, cnt.value)
            self.order.append(name)
            self.plist.append(p)
        self.packedT = torch.zeros(int(lib.mpdx_train_dgrad_pack_floats(h)), dtype=torch.float32, device=dev)
        # where the three parameters of the per-step checks live (module, attribute): named_parameters() walks the whole module tree - 0.15 ms per call,
        # twice per step it was most of the HOST time of an iteration (0.455 ms against 0.48 ms of GPU time at batch 32: round 6, tools/train_enqueue_probe2.py)
        self._spot = []
        for k in (self.order[0], self.order[len(self.order) // 2], self.order[-1]):
            parent, _, attr = k.rpartition(".")
            self._spot.append((k, unet.get_submodule(parent) if parent else unet, attr))
        self._n_checks = 0
        self._pending = None   # weakref to the _GradHolder of the autograd loss whose gradient currently sits in self.grad (snapshot_pending)

    def snapshot_pending(self):
        """The flat gradient buffer is shared by EVERY native pass on this U-Net (any TrainStep: trainer.train()'s own and the one behind
        model.loss()).  An autograd loss (_PLossesFn) normally reads it in place in its backward(); only when another pass is about
        to overwrite it first (summed losses, gradient accumulation, a second model.loss() before backward(), a TrainStep.step()) is
        that loss's gradient copied out - lazily, here, by whoever is about to overwrite the buffer.  Never called inside a graph capture."""
        ref = self._pending
        holder = ref() if ref is not None else None
        if holder is not None and holder.flat is None:
            holder.flat = self.grad.clone()
        self._pending = None

    def aliased(self, full: bool = False) -> bool:
        """Do the module's parameters still live in the flat vector?  (`.to()` / `.cuda()` / re-creating parameters breaks the
        aliasing.)  The per-step check looks at three parameters; `full` at all of them."""
        base = self.flat.data_ptr()
        self._n_checks += 1
        if not full and self._n_checks % 256:   # the spot check: three parameters looked up where they were registered (every 256th call walks the tree: a
            # replaced SUBMODULE keeps the old module object alive here)
            for k, mod, attr in self._spot:
                p = mod._parameters.get(attr)
                if p is None or p.data_ptr() != base + 4 * self.slices[k][0]:
                    return False
            return True
        named = dict(self.unet.named_parameters())
        keys = self.order if full else (self.order[0], self.order[len(self.order) // 2], self.order[-1])
        return all(k in named and named[k].data_ptr() == base + 4 * self.slices[k][0] for k in keys)

    def grads_bound(self, full: bool = False) -> bool:
        """Is every parameter's .grad still its view of the flat gradient?  (optimizer.zero_grad(set_to_none=True), the autograd
        bridge of model.loss() and `p.grad = None` all detach it.)"""
        base = self.grad.data_ptr()
        if not full:   # (the spot check, as in aliased())
            for k, mod, attr in self._spot:
                p = mod._parameters.get(attr)
                if p is None or p.grad is None or p.grad.data_ptr() != base + 4 * self.slices[k][0]:
                    return False
            return True
        named = dict(self.unet.named_parameters())
        return all(named[k].grad is not None and named[k].grad.data_ptr() == base + 4 * self.slices[k][0] for k in self.order)

    def bind_grads(self):
        """Re-point every p.grad at its slice of the flat gradient (what the native backward pass writes): torch optimisers and
        clip_grad_norm_ read p.grad, so a detached .grad would make them skip the parameter or apply a stale gradient silently."""
        named = dict(self.unet.named_parameters())
        for k in self.order:
            off, cnt = self.slices[k]
            p = named[k]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + cnt].view(p.shape)


def flat_params(unet) -> FlatParams:
    fp = getattr(unet, "_flat_params", None)
    if fp is None or not fp.aliased(full=True):
        fp = FlatParams(unet)
        unet._flat_params = fp
    return fp


class TrainStep:
    """loss + gradient of GaussianDiffusionModel.loss (diffusion_model_base.py:331-357) and the optimiser step, native."""

    def __init__(self, model):
        self.model = model
        self.unet = model.model
        self.fp = flat_params(self.unet)
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.scratch = torch.zeros(2048, dtype=torch.float32, device=self.fp.flat.device)
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=self.fp.flat.device)
        self.step_count = 0
        self._ws = None
        self._ws_B = 0
        half = 16
        self._freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device=self.fp.flat.device, dtype=torch.float32)

    def _packed(self):
        lib, h = _lib.load(), self.unet._handle()
        u = self.unet
        if u._packed is None or u._packed.device != self.fp.flat.device:
            u._packed = torch.zeros(lib.mpdx_unet_packed_floats(h), dtype=torch.float32, device=self.fp.flat.device)
        return u._packed

    def pack(self, sync_engine: bool = True):
        """flat parameters -> the two kernel layouts (after every optimiser step).  `sync_engine`: also tell the TemporalUnet's
        inference engine that its pack is current (a walk over the parameters; the per-step call inside loss_backward skips it and
        leaves the engine marked stale instead)."""
        if not self.fp.aliased():
            raise RuntimeError("the model's parameters no longer alias the flat training vector (was the model moved or re-created?) - "
                               "build a new TrainStep")
        lib, h = _lib.load(), self.unet._handle()
        packed = self._packed()
        _lib.check(lib.mpdx_train_pack(h, self.fp.flat.data_ptr(), packed.data_ptr(), self.fp.packedT.data_ptr(), _lib.current_stream()),
                   "mpdx_train_pack")
        # the inference engine of this TemporalUnet sees the new weights: its pack is current, its time table is not
        self.unet._stamp = self.unet._param_stamp() if sync_engine else None
        self.unet._timetab, self.unet._timetab_T = None, 0

    def loss_backward(self, x_start, hard_conds=None, t=None, noise=None, loss_scale=1.0, bind_grads=True, _static_loss=False):
        """(loss, info) as model.loss(x, None, hard_conds) returns them, with d loss / d parameters left in every p.grad
        (overwritten, not accumulated - the reference zeroes the gradients before every backward, trainer.py:262-263)."""
        m = self.model
        if not x_start.is_cuda:
            raise RuntimeError("training runs on the GPU (libmpdx); there is no CPU fallback")
        if m.loss_type not in ("l1", "l2"):
            raise NotImplementedError(m.loss_type)
        lib, h = _lib.load(), self.unet._handle()
        x_start = x_start.to(torch.float32).contiguous()
        B, H, D = x_start.shape
        dev = x_start.device
        if t is None:
            t = torch.randint(0, m.n_diffusion_steps, (B,), device=dev).long()   # diffusion_model_base.py:356
        t = t.to(device=dev, dtype=torch.long).reshape(-1).contiguous()
        if noise is None:
            noise = m.fill_randn(torch.empty((B, H, D), device=dev, dtype=torch.float32))
        noise = noise.to(torch.float32).contiguous()
        hs, hg = m._hard_tables(hard_conds, B, H, D, dev)
        if self._ws is None or self._ws_B < B:
            self._ws = torch.empty(int(lib.mpdx_train_workspace_floats(h, B)), dtype=torch.float32, device=dev)
            self._ws_B = B
            # graphs captured by step() hold the OLD workspace's address: drop them (they are re-captured after their next eager calls);
            # with no graph captured yet there is nothing to invalidate, and the warm-up count of the running sequence stands
            if self.__dict__.pop("_graphs", None):
                self.__dict__.pop("_graph_state", None)
        self.pack(sync_engine=False)
        if not torch.cuda.is_current_stream_capturing():   # (a capture must not record the copy; step() snapshots before it captures / replays)
            self.fp.snapshot_pending()   # an autograd loss whose backward() has not run yet still needs the gradients about to be overwritten
        _lib.check(lib.mpdx_train_loss_backward(
            h, self.fp.flat.data_ptr(), self._packed().data_ptr(), self.fp.packedT.data_ptr(), self.fp.grad.data_ptr(), x_start.data_ptr(),
            noise.data_ptr(), t.data_ptr(), m.sqrt_alphas_cumprod.data_ptr(), m.sqrt_one_minus_alphas_cumprod.data_ptr(),
            self._freqs.data_ptr(), hs.data_ptr() if hs is not None else None, hg.data_ptr() if hg is not None else None, None,
            int(m.n_diffusion_steps), B, 1 if m.predict_epsilon else 0, 1 if m.loss_type == "l1" else 0, float(loss_scale),
            self.loss_buf.data_ptr(), self._ws.data_ptr(), _lib.current_stream()), "mpdx_train_loss_backward")
        if bind_grads and not self.fp.grads_bound():   # the docstring's promise: the gradients ARE in p.grad after this call
            self.fp.bind_grads()
        return (self.loss_buf[0] if _static_loss else self.loss_buf[0].clone()), {}   # (_static_loss: the graph's own output tensor, TrainStep.step)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=None):
        """clip_grad_norm_(max_norm) if given, then one torch.optim.Adam step (defaults as trainer.py:140); returns the
        0-dim tensor holding the gradient norm before clipping (or None)."""
        lib = _lib.load()
        self.step_count += 1
        mn = float(max_norm) if max_norm else 0.0
        _lib.check(lib.mpdx_adam_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      self.fp.n, float(lr), float(betas[0]), float(betas[1]), float(eps), self.step_count, mn,
                                      self.scratch.data_ptr(), _lib.current_stream()), "mpdx_adam_step")
        self.unet._stamp = None   # the inference engine's pack of these weights is stale until the next pack()
        return self.scratch[0] if mn > 0 else None

    # ------------------------------------------------------------------------------------------------ one iteration, eager or as a hipGraph
    _MAX_GRAPHS = 4   # captured iterations kept (least recently used beyond that are dropped: each owns a private memory pool)
    _MAX_STATES = 16  # (shapes, hyper-parameters) whose launch form is remembered

    def step(self, x_start, hard_conds=None, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=None, t=None, noise=None, use_graph=None):
        """One training iteration of trainer.py:186-283 - p_losses + backward (loss_backward) and clip + Adam - returning the loss.

        The iteration runs as eager launches or is REPLAYED AS ONE hipGraph (torch.cuda.CUDAGraph over the same native launches): at the reference's
        batch of 32 the launches of an iteration take the host as long to enqueue as the GPU to run, and a slower host is then what a step costs;
        the replay is one host call.  Whether that pays depends on the box, so by default it is MEASURED per (shapes, hyper-parameters): calls 1-3
        run eager (2 and 3 timed), call 4 captures, calls 5-6 time the replay, and the faster form is kept (`launch_mode()` says which).
        use_graph=True / False (or MPDX_TRAIN_GRAPH=1 / 0) force either.

        BOTH forms are the same launches with the same arguments, so the choice never changes the numbers: t and the noise are drawn ON THE DEVICE by
        the pass's first launch (mpdx_train_draw: Philox keyed by the model's seed and the device-resident step count - what
        diffusion_model_base.py:356 / :337 draw with torch.randint / torch.randn_like; pass `t` / `noise` to supply them instead), Adam's step count
        and the learning rate live on the device (mpdx_adam_step with step < 0, lr < 0: scratch[4], scratch[5]) - a fixed seed gives the same
        losses and weights whichever form a host ends up with, and an LR schedule neither re-captures a graph nor grows any cache.
        In graph mode the batch and the hard conditions are copied into the graph's static buffers, and the returned 0-dim tensor is the graph's
        own output buffer: the next replay overwrites it - read it (float(loss)) or clone it before the next call if you keep it (the eager form
        returns a fresh tensor)."""
        import os
        import time
        env = os.environ.get("MPDX_TRAIN_GRAPH")
        if use_graph is None and env is not None:
            use_graph = env != "0"
        hard_conds = hard_conds or {}
        mn = float(max_norm) if max_norm else 0.0
        key = (tuple(x_start.shape), tuple(sorted((int(k), tuple(v.shape)) for k, v in hard_conds.items())), tuple(betas), float(eps), mn,
               t is not None, noise is not None)
        graphs = self.__dict__.setdefault("_graphs", {})
        state = self.__dict__.setdefault("_graph_state", {})
        s = state.pop(key, None) or {"calls": 0, "eager_ms": [], "graph_ms": [], "mode": None, "bufs": None}   # mode: None undecided, "graph", "eager"
        state[key] = s   # most recently used last
        while len(state) > self._MAX_STATES:
            old = next(iter(state))
            state.pop(old)
            graphs.pop(old, None)
        if use_graph is not None:
            s["mode"] = "graph" if use_graph else "eager"
        g = graphs.get(key) if s["mode"] != "eager" else None
        if g is not None and g["ptrs"] != self._static_ptrs():
            # a buffer the graph reads or writes has moved since the capture (e.g. the TemporalUnet's inference engine re-created its weight pack after
            # a summary / validation pass): the graph is stale - drop it; it is re-captured after the next eager steps
            del graphs[key]
            s["calls"], g = 0, None
        s["calls"] += 1
        if not self.fp.aliased():
            raise RuntimeError("the model's parameters no longer alias the flat training vector (was the model moved or re-created?) - build a new TrainStep")
        # device-resident step count and learning rate (both forms read them there)
        if self.__dict__.get("_dev_steps") != self.step_count:   # adam_step() calls in between moved the host's count: re-seed the device's
            self.scratch.view(torch.int32)[4] = self.step_count
        if self.__dict__.get("_dev_lr") != float(lr):
            self.scratch[5:6].fill_(float(lr))
            self._dev_lr = float(lr)
        timed = None
        if g is None:
            if s["mode"] == "eager" or s["calls"] < (4 if s["mode"] is None else 3):   # eager (also the warm-up of everything a capture must not do:
                # allocations, one-off attribute calls); a forced graph is captured on the third call, a measured one on the fourth
                measure = s["mode"] is None and s["calls"] in (2, 3)
                if measure:
                    torch.cuda.synchronize()
                    timed = time.perf_counter()
                draw = t is None and noise is None
                if draw:
                    if s["bufs"] is None or s["bufs"][1].device != x_start.device:
                        s["bufs"] = (torch.zeros(x_start.shape[0], dtype=torch.long, device=x_start.device),
                                     torch.empty(tuple(x_start.shape), dtype=torch.float32, device=x_start.device))
                    tt, nz = s["bufs"]
                else:   # one of the two supplied: the other from torch's generator, as inside a capture
                    tt = t if t is not None else torch.randint(0, self.model.n_diffusion_steps, (x_start.shape[0],), device=x_start.device).long()
                    nz = noise if noise is not None else torch.randn_like(x_start, dtype=torch.float32)
                self.fp.snapshot_pending()
                loss = self._iteration(x_start, hard_conds, tt, nz, draw, betas, eps, mn, static_loss=False)
                self._after_iteration()
                if measure:
                    torch.cuda.synchronize()
                    s["eager_ms"].append((time.perf_counter() - timed) * 1e3)
                return loss
            self.fp.snapshot_pending()   # (outside the capture)
            g = graphs[key] = self._capture(x_start, hard_conds, betas, eps, mn, t, noise)
            while len(graphs) > self._MAX_GRAPHS:
                graphs.pop(next(iter(graphs)))
        else:
            graphs[key] = graphs.pop(key)   # most recently used last
        measure = s["mode"] is None and len(s["graph_ms"]) < 2 and s["calls"] > 4
        if measure:
            torch.cuda.synchronize()
            timed = time.perf_counter()
        self.fp.snapshot_pending()   # the replay overwrites the flat gradient: a pending autograd loss keeps its own copy
        dsts, srcs = [g["x"]] + [g["hc"][k] for k in hard_conds], [x_start] + [hard_conds[k] for k in hard_conds]
        if noise is not None:
            dsts.append(g["noise"]); srcs.append(noise)
        # the batch (and a supplied noise tensor) by plain copies - contiguous device-to-device: the runtime's copy kernel, 2-3 us; the hard conditions
        # (a few KB each) in ONE launch.  (All of them in one _foreach_copy_ was 4.7 us at batch 32 but 17.5 us at batch 128 x D = 14: multi_tensor_apply
        # hands a 458-KB tensor to two blocks - profiles/r05_train128_kernel_stats.csv.)
        # (round 6: a LARGE contiguous tensor goes into the same launch as 16 K-element slices - multi_tensor_apply spreads list entries over blocks,
        #  so the batch of 128 x 64 x 14 floats is seven entries instead of a copy launch of its own: one launch for all inputs at every batch size)
        small_d, small_s = [], []
        for a, b in zip(dsts, srcs):
            if a.dtype == b.dtype and a.device == b.device and a.shape == b.shape and a.numel() <= 16384:
                small_d.append(a); small_s.append(b)
            elif a.dtype == b.dtype and a.device == b.device and a.shape == b.shape and b.is_contiguous() and a.is_contiguous() and a.numel() <= (1 << 20):
                small_d.extend(a.view(-1).split(16384)); small_s.extend(b.view(-1).split(16384))
            else:
                a.copy_(b, non_blocking=True)
        if small_d:
            torch._foreach_copy_(small_d, small_s)
        if t is not None:
            g["t"].copy_(t, non_blocking=True)
        g["graph"].replay()
        self._after_iteration()
        if measure:
            torch.cuda.synchronize()
            s["graph_ms"].append((time.perf_counter() - timed) * 1e3)
            if len(s["graph_ms"]) == 2:   # both forms measured on THIS host and GPU: keep the faster (ties go to the graph: one host call per step)
                s["mode"] = "graph" if min(s["graph_ms"]) <= 1.02 * min(s["eager_ms"] or [float("inf")]) else "eager"
                if s["mode"] == "eager":
                    loss = g["loss"].clone()
                    del graphs[key]
                    return loss
        return g["loss"]

    def _iteration(self, x, hard_conds, tt, nz, draw, betas, eps, mn, static_loss):
        """the launches of one iteration - what step() runs eagerly and what _capture records: loss_backward (with the device draw of t and the noise
        armed when `draw`) and clip + Adam reading the step count and the learning rate from the device"""
        lib, m = _lib.load(), self.model
        if draw:
            _lib.check(lib.mpdx_train_draw(self.unet._handle(), int(m._rng_seed) & (2 ** 64 - 1), self.scratch.data_ptr() + 16), "mpdx_train_draw")
        try:
            loss, _ = self.loss_backward(x, hard_conds, t=tt, noise=nz, _static_loss=static_loss)
        finally:
            if draw:
                lib.mpdx_train_draw(self.unet._handle(), 0, None)
        _lib.check(lib.mpdx_adam_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      self.fp.n, -1.0, float(betas[0]), float(betas[1]), float(eps), -1, mn,
                                      self.scratch.data_ptr(), _lib.current_stream()), "mpdx_adam_step")
        return loss

    def _after_iteration(self):
        self.step_count += 1
        self._dev_steps = self.step_count
        self.unet._stamp = None   # the inference engine's pack of these weights is stale until the next pack()
        self.unet._timetab, self.unet._timetab_T = None, 0
        if not self.fp.grads_bound():
            self.fp.bind_grads()

    def launch_mode(self):
        """how step() runs each (shapes, hyper-parameters) it has seen: {'mode': 'graph' | 'eager' | None (still measuring), 'eager_ms', 'graph_ms'}"""
        return [{"batch": k[0][0], "mode": v["mode"], "eager_ms": [round(x, 3) for x in v["eager_ms"]], "graph_ms": [round(x, 3) for x in v["graph_ms"]]}
                for k, v in self.__dict__.get("_graph_state", {}).items()]

    def _static_ptrs(self):
        """addresses of every persistent buffer a captured iteration touches"""
        return (self._packed().data_ptr(), self.fp.packedT.data_ptr(), self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.exp_avg.data_ptr(),
                self.exp_avg_sq.data_ptr(), self.scratch.data_ptr(), self.loss_buf.data_ptr(), 0 if self._ws is None else self._ws.data_ptr())

    def _capture(self, x_start, hard_conds, betas, eps, mn, t, noise):
        dev = x_start.device
        B = x_start.shape[0]
        st = {"x": x_start.to(torch.float32).contiguous().clone(), "hc": {k: v.to(device=dev, dtype=torch.float32).contiguous().clone() for k, v in hard_conds.items()},
              "t": None if t is None else t.to(device=dev, dtype=torch.long).reshape(-1).contiguous().clone(),
              "noise": None if noise is None else noise.to(torch.float32).contiguous().clone()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        draw = st["t"] is None and st["noise"] is None   # both drawn: on the device, inside the pass's first launch (mpdx_train_draw)
        if draw:
            st["t"], st["noise"] = torch.zeros(B, dtype=torch.long, device=dev), torch.empty_like(st["x"])
        m = self.model
        with torch.cuda.graph(graph, stream=side):
            tt = st["t"] if st["t"] is not None else torch.randint(0, m.n_diffusion_steps, (B,), device=dev).long()
            nz = st["noise"] if st["noise"] is not None else torch.randn_like(st["x"])
            loss = self._iteration(st["x"], st["hc"], tt, nz, draw, betas, eps, mn, static_loss=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        st["graph"], st["loss"], st["ptrs"] = graph, loss, self._static_ptrs()
        return st


class _PLossesFn(torch.autograd.Function):
    """autograd bridge: forward = TrainStep.loss_backward (loss AND gradients in one native pass), backward hands each parameter its
    slice of the flat gradient times the incoming gradient - so the reference's own loop body
    `loss, info = model.loss(x, context, hard_conds); loss.backward(); optimizer.step()` runs unchanged."""

    @staticmethod
    def forward(ctx, step, x_start, hard_conds, t, noise, *params):
        loss, _ = step.loss_backward(x_start, hard_conds, t=t, noise=noise, bind_grads=False)
        ctx.step = step
        ctx.n_params = len(params)
        # THIS call's gradients sit in the shared flat buffer; they are copied out only if another native pass runs before this
        # loss's backward() (FlatParams.snapshot_pending) - the common single-loss iteration makes no parameter-sized copy
        ctx.holder = _GradHolder()
        step.fp._pending = weakref.ref(ctx.holder)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        fp = ctx.step.fp
        flat = ctx.holder.flat if ctx.holder.flat is not None else fp.grad
        scaled = flat * grad_out   # ONE parameter-sized product; the per-parameter gradients are views of it
        grads = []
        for name, p in zip(fp.order, fp.plist):
            off, cnt = fp.slices[name]
            grads.append(scaled[off:off + cnt].view(p.shape))
        return (None, None, None, None, None) + tuple(grads)


class _GradHolder:
    """weakly referenced by FlatParams._pending: `flat` is filled only when the shared gradient buffer is about to be overwritten"""
    __slots__ = ("flat", "__weakref__")

    def __init__(self):
        self.flat = None


def loss_with_grad(model, x_start, hard_conds=None, t=None, noise=None):
    """model.loss(x, None, hard_conds) as a 0-dim tensor WITH autograd history (its backward() fills / accumulates p.grad of every
    TemporalUnet parameter, as torch autograd would).  GaussianDiffusionModel.loss returns this when gradients are enabled."""
    step = getattr(model, "_train_step", None)
    if step is None or not step.fp.aliased():
        step = TrainStep(model)
        model._train_step = step
    params = step.fp.plist   # (the Parameter objects FlatParams re-pointed at the flat vector: aliased() above vouches for them)
    # p.grad aliases the flat gradient that the native pass overwrites: autograd must accumulate into its own tensors
    g0 = step.fp.grad.data_ptr()
    g1 = g0 + 4 * step.fp.n
    for p in params:
        g = p.grad
        if g is not None and g0 <= g.data_ptr() < g1:
            p.grad = None
    return _PLossesFn.apply(step, x_start, hard_conds, t, noise, *params)


class EMA:
    """trainer.py:67-85.  update_model_average takes two models like the reference; when both are flat (FlatParams) it is
    one kernel over the flat vectors."""

    def __init__(self, beta=0.995):
        self.beta = beta

    def update_model_average(self, ema_model, current_model):
        eu, cu = getattr(ema_model, "model", ema_model), getattr(current_model, "model", current_model)
        ef, cf = flat_params(eu), flat_params(cu)
        _lib.check(_lib.load().mpdx_ema_update(ef.flat.data_ptr(), cf.flat.data_ptr(), ef.n, float(self.beta), _lib.current_stream()),
                   "mpdx_ema_update")
        eu._stamp = None   # the EMA model's inference pack is stale

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


class GaussianDiffusionLoss:
    """mpd/losses/gaussian_diffusion_loss.py:6-28 - forward value (no autograd history), used for validation."""

    @staticmethod
    def loss_fn(diffusion_model, input_dict, dataset, step=None):
        traj_normalized = input_dict[f"{dataset.field_key_traj}_normalized"]
        hard_conds = input_dict.get("hard_conds", {})
        loss, info = diffusion_model.loss(traj_normalized, None, hard_conds)
        return {"diffusion_loss": loss}, info


def save_model_to_disk(model, epoch, total_steps, checkpoints_dir=None, prefix="model_"):   # trainer.py:29-37
    if hasattr(model, "is_frozen") and model.is_frozen:
        return
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(checkpoints_dir, f"{prefix}current_state_dict.pth"))
    torch.save(sd, os.path.join(checkpoints_dir, f"{prefix}epoch_{epoch:04d}_iter_{total_steps:06d}_state_dict.pth"))


def save_models_to_disk(models_prefix_l, epoch, total_steps, checkpoints_dir=None):   # trainer.py:20-26
    for model, prefix in models_prefix_l:
        if model is not None:
            save_model_to_disk(model, epoch, total_steps, checkpoints_dir, prefix=f"{prefix}_")


def save_losses_to_disk(train_losses, val_losses, checkpoints_dir=None):   # trainer.py:40-42
    np.save(os.path.join(checkpoints_dir, "train_losses.npy"), np.array(train_losses, dtype=object), allow_pickle=True)
    np.save(os.path.join(checkpoints_dir, "val_losses.npy"), np.array(val_losses, dtype=object), allow_pickle=True)


def train(model=None, train_dataloader=None, epochs=None, lr=None, steps_til_summary=None, model_dir=None, loss_fn=None,
          train_subset=None, summary_fn=None, steps_til_checkpoint=None, val_dataloader=None, val_subset=None, clip_grad=False,
          clip_grad_max_norm=1.0, val_loss_fn=None, optimizers=None, steps_per_validation=10, max_steps=None, use_ema: bool = True,
          ema_decay: float = 0.995, step_start_ema: int = 1000, update_ema_every: int = 10, use_amp=False, early_stopper_patience=-1,
          debug=False, tensor_args=None, **kwargs):
    """trainer.py:120-320 with the same arguments.  Per step: loss + gradients (TrainStep.loss_backward), optional
    clip_grad_norm_, Adam (native unless `optimizers` - torch optimisers over model.parameters() - are given), EMA every
    `update_ema_every` steps (reset to the model before `step_start_ema`).  Differences, all outside the compute path: no wandb
    (not installed here), no AMP (`use_amp=True` raises: the kernels are fp32 like the reference's default), checkpoints hold
    state dicts only.  A `loss_fn` other than GaussianDiffusionLoss.loss_fn (the reference's hook: loss_fn(model, batch, dataset) ->
    ({name: loss}, info)) is honoured the reference's way: its losses are summed, `.backward()` runs through the autograd bridge of
    `model.loss` and torch optimisers step (created as torch.optim.Adam(lr) when none are passed).  Returns (model, ema_model, train_losses)."""
    if use_amp:
        raise NotImplementedError("use_amp=True: the training kernels are fp32 (the reference's default is use_amp=False)")
    if model is None or train_dataloader is None or epochs is None:
        raise ValueError("model, train_dataloader and epochs are required")
    field = train_subset.dataset.field_key_traj if train_subset is not None else "traj"
    ema_model = None
    if use_ema:
        ema = EMA(beta=ema_decay)
        ema_model = copy.deepcopy(model)
    step_fn = TrainStep(model)
    custom_loss = loss_fn is not None and loss_fn is not GaussianDiffusionLoss.loss_fn
    if custom_loss and optimizers is None:
        optimizers = [torch.optim.Adam(lr=lr, params=model.parameters())]   # trainer.py:140
    if val_dataloader is not None and val_loss_fn is None:
        raise AssertionError("If validation set is passed, have to pass a validation loss_fn!")
    checkpoints_dir = None
    if model_dir is not None:
        os.makedirs(model_dir, exist_ok=True)
        os.makedirs(os.path.join(model_dir, "summaries"), exist_ok=True)
        checkpoints_dir = os.path.join(model_dir, "checkpoints")
        os.makedirs(checkpoints_dir, exist_ok=True)
        save_models_to_disk([(model, "model"), (ema_model, "ema_model")], 0, 0, checkpoints_dir)
    max_norm = None
    if clip_grad:
        max_norm = clip_grad_max_norm if isinstance(clip_grad, bool) else clip_grad
    train_steps_current = 0
    train_losses_l, validation_losses_l = [], []
    early_stopper = EarlyStopper(patience=early_stopper_patience, min_delta=0)   # trainer.py:161
    stop_training = False
    total_val_loss = None   # no validation has run yet (the reference's variable does not exist until then)
    dev = next(model.parameters()).device

    def ema_update():
        if train_steps_current < step_start_ema:
            flat_params(ema_model.model).flat.copy_(flat_params(model.model).flat)   # ema_model.load_state_dict(model.state_dict())
            ema_model.model._stamp = None
        ema.update_model_average(ema_model, model)

    for epoch in range(epochs):
        model.train()
        for step, batch in enumerate(train_dataloader):
            fused_step = False
            if custom_loss:   # the reference's generic path (trainer.py:186-197, 262-266)
                bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
                if "hard_conds" in bd:
                    bd["hard_conds"] = {k: v.to(dev) for k, v in bd["hard_conds"].items()}
                losses, info = loss_fn(model, bd, train_subset.dataset if train_subset is not None else None)
                loss = sum(l.mean() for l in losses.values())
                for opt in optimizers:
                    opt.zero_grad()
                loss.backward()
                loss = loss.detach()
            else:
                x = batch[f"{field}_normalized"].to(dev)
                hard_conds = {k: v.to(dev) for k, v in batch.get("hard_conds", {}).items()}
                # the native optimiser: loss + backward + clip + Adam as ONE call (replayed as a hipGraph, TrainStep.step); on summary steps the two
                # halves run apart, because the summary / validation below look at the model BEFORE the optimiser step (trainer.py:199-266)
                fused_step = optimizers is None and not (steps_til_summary and train_steps_current % steps_til_summary == 0)
                if fused_step:
                    loss, info = step_fn.step(x, hard_conds, lr, max_norm=max_norm), {}
                else:
                    loss, info = step_fn.loss_backward(x, hard_conds)
            if steps_til_summary and train_steps_current % steps_til_summary == 0:
                lv = float(loss)   # the only host synchronisation of a step, on summary steps
                train_losses_l.append((train_steps_current, {"diffusion_loss": lv}))
                print(f"train_steps_current: {train_steps_current}  diffusion_loss {lv:.6f}")
                if summary_fn is not None:   # do_summary (trainer.py:88-113): no_grad, eval() around the call, train() after it
                    sm = ema_model if ema_model is not None else model
                    with torch.no_grad():
                        sm.eval()
                        summary_fn(train_steps_current, sm, batch_dict=batch, loss_info=info,
                                   datasubset=train_subset, prefix="TRAINING ", debug=debug, tensor_args=tensor_args)
                    sm.train()
                if val_dataloader is not None:
                    vals = []
                    for step_val, vb in enumerate(val_dataloader):
                        vb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in vb.items()}
                        if "hard_conds" in vb:
                            vb["hard_conds"] = {k: v.to(dev) for k, v in vb["hard_conds"].items()}
                        with torch.no_grad():   # forward value only (the reference runs validation without stepping the optimiser)
                            vl, _ = val_loss_fn(model, vb, val_subset.dataset, step=train_steps_current)
                        vals.append(float(sum(v.mean() for v in vl.values())))
                        if step_val == steps_per_validation:
                            break
                    validation_losses_l.append((train_steps_current, {"VALIDATION diffusion_loss": float(np.mean(vals))}))
                    total_val_loss = float(np.sum(vals))   # trainer.py:225-233
            # trainer.py:269 evaluates the stopper EVERY training step with the last validation total (patience counts steps)
            if total_val_loss is not None and early_stopper.early_stop(total_val_loss):
                print(f"Early stopped training at {train_steps_current} steps.")
                stop_training = True
            if optimizers is None:
                if not fused_step:
                    step_fn.adam_step(lr, max_norm=max_norm)
            else:   # torch optimisers over the same (aliased) parameters, as the reference runs them
                if not custom_loss:
                    # the native pass wrote the flat gradient; an earlier model.loss() with autograd or zero_grad(set_to_none=True)
                    # may have detached p.grad from it - without this the optimiser would skip every parameter, silently
                    if not step_fn.fp.aliased(full=True):
                        raise RuntimeError("the model's parameters no longer alias the flat training vector - build a new TrainStep")
                    step_fn.fp.bind_grads()
                if max_norm is not None:
                    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
                for opt in optimizers:
                    opt.step()
            if ema_model is not None and train_steps_current % update_ema_every == 0:
                ema_update()
            train_steps_current += 1
            if checkpoints_dir is not None and steps_til_checkpoint is not None and train_steps_current % steps_til_checkpoint == 0:
                step_fn.pack()
                save_models_to_disk([(model, "model"), (ema_model, "ema_model")], epoch, train_steps_current, checkpoints_dir)
                save_losses_to_disk(train_losses_l, validation_losses_l, checkpoints_dir)
            if stop_training or (max_steps is not None and train_steps_current == max_steps):
                break
        if stop_training or (max_steps is not None and train_steps_current == max_steps):
            break
    if ema_model is not None:
        ema_update()
    step_fn.pack()
    if checkpoints_dir is not None:
        save_models_to_disk([(model, "model"), (ema_model, "ema_model")], epoch, train_steps_current, checkpoints_dir)
        save_losses_to_disk(train_losses_l, validation_losses_l, checkpoints_dir)
    return model, ema_model, train_losses_l
