"""Sharding of independent start/goal contexts across the GPUs of one node (BASELINE configs[4]).

The reference has no distributed code at all.  Contexts (start/goal pairs) are independent planning problems, and
GroupNorm normalises per trajectory, so the batch shards with ZERO exchange during the loop: rank r plans a contiguous
block of contexts with replicated weights/obstacles.  The only collective is the final gather of the planned
trajectories (RCCL all-gather over xGMI: each rank's shard travels once over its direct link; for the CPU tests the
same code runs on the gloo backend).  One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) block of `n_items` for `rank` (first n_items % world_size ranks get one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def expand_contexts(start: torch.Tensor, goal: torch.Tensor, n_samples: int):
    """[C,D] start/goal tables -> per-trajectory hard-condition tables [C*n_samples, D] (the reference's
    einops 'd -> b d' repeat, diffusion_model_base.py:292-294, generalised to several contexts)."""
    return start.repeat_interleave(n_samples, dim=0).contiguous(), goal.repeat_interleave(n_samples, dim=0).contiguous()


def plan_contexts(model, start: torch.Tensor, goal: torch.Tensor, n_samples: int, *, rank: int = 0, world_size: int = 1,
                  max_batch: int = 8192, planner=None, **plan_kwargs) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Plan this rank's block of contexts.  start/goal: normalised [C_total, D] tables (identical on every rank).
    Returns (final trajectories [C_local*n_samples, H, D], (lo, hi) context range).  Contexts are processed in chunks of
    at most `max_batch` trajectories; the whole-tensor range test of the normaliser is evaluated per context
    (n_per_context=n_samples), as one reference run_inference call per context would.
    `planner(hard_conds, B, n_per_context, **kw) -> x` defaults to model.plan (tests inject a stub)."""
    C_total = start.shape[0]
    lo, hi = shard_range(C_total, world_size, rank)
    H = plan_kwargs.pop("horizon", None)
    outs = []
    per_chunk = max(1, max_batch // n_samples)
    if planner is None:
        def planner(hc, B, npc, **kw):
            x, _ = model.plan(hc, B, H, return_chain=False, n_per_context=npc, **kw)
            return x
    for c0 in range(lo, hi, per_chunk):
        c1 = min(hi, c0 + per_chunk)
        hs, hg = expand_contexts(start[c0:c1], goal[c0:c1], n_samples)
        horizon_idx = (H or getattr(getattr(model, "model", None), "n_support_points", 64)) - 1
        outs.append(planner({0: hs, horizon_idx: hg}, (c1 - c0) * n_samples, n_samples, **plan_kwargs))
    if not outs:
        D = start.shape[1]
        return start.new_zeros((0, (H or 64), D)), (lo, hi)
    return torch.cat(outs, dim=0), (lo, hi)


def gather_trajectories(local: torch.Tensor, n_contexts: int, n_samples: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """All-gather the per-rank trajectory blocks into [n_contexts*n_samples, H, D] on every rank (one collective).
    Blocks may differ by one context in size: they are padded to the largest block for the collective.
    force_collective: issue the all-gather even in a world of one rank (exercises the RCCL path on a single GPU)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    if dist.get_world_size(group) == 1 and not force_collective:
        return local
    world = dist.get_world_size(group)
    sizes = [(shard_range(n_contexts, world, r)[1] - shard_range(n_contexts, world, r)[0]) * n_samples for r in range(world)]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves host memory: stage through the CPU (the CPU tests' backend, and two ranks sharing one GPU in the GPU test;
        # production runs one rank per GPU on the nccl = RCCL backend, device to device)
        host = pad.contiguous().cpu()
        out_h = host.new_empty((world * mx,) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out_h, host, group=group)
        out = out_h.to(local.device)
    else:
        out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
