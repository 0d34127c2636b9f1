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


def block_sizes(n_contexts: int, n_samples: int, world: int):
    """Trajectories per rank block (shard_range of the contexts x n_samples)."""
    return [(shard_range(n_contexts, world, r)[1] - shard_range(n_contexts, world, r)[0]) * n_samples for r in range(world)]


_CK_MULT = 0x9E3779B1   # odd 32-bit constant: the position weight (index * _CK_MULT + 1) is odd, hence never zero mod 2^64


def shard_checksum(x: torch.Tensor) -> torch.Tensor:
    """POSITION-WEIGHTED 64-bit checksum of a float32 tensor's BIT PATTERNS: sum_i bits_i * (i * odd_const + 1) + numel in
    wrapping int64 arithmetic.  A transport check for the gather: a flipped bit, a truncated block, a block in the wrong slot,
    and - unlike a plain sum - trajectories permuted inside a block or two compensating bit errors at different positions all
    change it (a detection code, not a cryptographic one)."""
    return block_checksums(x, [x.shape[0] if x.dim() else x.numel()])[0]


def block_checksums(x: torch.Tensor, sizes) -> torch.Tensor:
    """shard_checksum of every consecutive block of `sizes[r]` leading-dimension entries of x, as one int64 vector computed by
    a handful of tensor ops (positions are block-local, so a block's checksum does not depend on where it sits in x)."""
    world = len(sizes)
    out = torch.zeros(world, dtype=torch.int64, device=x.device)
    if x.numel() == 0:
        return out
    per = x[0].numel() if x.dim() > 1 else 1
    bits = x.contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    counts = torch.tensor([int(s) * per for s in sizes], dtype=torch.int64, device=x.device)
    starts = torch.cumsum(counts, 0) - counts
    block = torch.repeat_interleave(torch.arange(world, device=x.device), counts, output_size=bits.numel())
    local = torch.arange(bits.numel(), dtype=torch.int64, device=x.device) - starts[block]
    out.index_add_(0, block, bits * (local * _CK_MULT + 1))
    return out + counts


def verify_gather(full: torch.Tensor, local: torch.Tensor, n_contexts: int, n_samples: int, group=None) -> bool:
    """Every rank publishes shard_checksum(local) (one 8-byte all-gather) and re-computes the checksum of every block of the
    gathered tensor (one batched tensor op, ONE host synchronisation): True when all blocks match what their owners planned.
    A world without a process group checks itself."""
    import torch.distributed as dist
    mine = shard_checksum(local).reshape(1)
    if not dist.is_available() or not dist.is_initialized():
        return bool(shard_checksum(full) == mine[0])
    world = dist.get_world_size(group)
    gloo = dist.get_backend(group) == "gloo"
    sums = [torch.zeros_like(mine.cpu() if gloo else mine) for _ in range(world)]
    dist.all_gather(sums, mine.cpu() if gloo else mine, group=group)
    sizes = block_sizes(n_contexts, n_samples, world)
    if full.shape[0] != sum(sizes):
        return False
    got = block_checksums(full, sizes)
    want = torch.cat([s.reshape(1) for s in sums]).to(got.device)
    return bool(torch.equal(got, want))


def gather_trajectories(local: torch.Tensor, n_contexts: int, n_samples: int, group=None, force_collective: bool = False,
                        mode: Optional[str] = None, timeout_s: Optional[float] = None) -> torch.Tensor:
    """Gather the per-rank trajectory blocks into [n_contexts*n_samples, H, D] on every rank - the path's ONE exchange step.

    mode "collective" (default; MPDX_GATHER=collective): one all_gather_into_tensor.  Blocks may differ by one context in size:
        they are padded to the largest block for the collective.
    mode "one_hop" (MPDX_GATHER=one_hop): world-1 isend + world-1 irecv per rank, issued as ONE batch (dist.batch_isend_irecv =
        one grouped RCCL launch): every shard travels exactly once over the DIRECT xGMI link to each peer and lands in its slot of
        the output (no padding, no ring: a ring all-gather forwards every block over world-1 hops and is bound by one link;
        MI355X links are point to point, 7 per GPU - SURVEY.md section 5).  Same result, bit for bit; bench.py times both.
    force_collective: take the collective path even in a world of one rank (exercises RCCL on a single GPU).
    timeout_s: bound every wait of the one-hop form (a peer that failed before posting its sends then raises here instead of
        blocking this rank forever); None = wait without a limit."""
    import os
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    if dist.get_world_size(group) == 1 and not force_collective:
        return local
    mode = mode or os.environ.get("MPDX_GATHER", "collective")
    if mode not in ("collective", "one_hop"):
        raise ValueError(f"gather mode {mode!r} (collective | one_hop)")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = block_sizes(n_contexts, n_samples, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} trajectories, its block of {n_contexts} contexts x {n_samples} has {sizes[rank]}")
    # gloo moves host memory: stage through the CPU (the CPU tests' backend, and several ranks sharing one GPU in the single-GPU
    # rig; production runs one rank per GPU on the nccl = RCCL backend, device to device)
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.contiguous().cpu() if via_host else local.contiguous()
    if mode == "one_hop" and world > 1:
        out = src.new_empty((sum(sizes),) + tuple(src.shape[1:]))
        offs = [sum(sizes[:r]) for r in range(world)]
        out[offs[rank]:offs[rank] + sizes[rank]] = src
        ops = []
        for k in range(1, world):   # peer order rotated by rank: at step k every rank sends to rank+k and receives from rank-k
            to, frm = (rank + k) % world, (rank - k) % world
            to_g = dist.get_global_rank(group, to) if group is not None else to
            frm_g = dist.get_global_rank(group, frm) if group is not None else frm
            if sizes[rank]:
                ops.append(dist.P2POp(dist.isend, src, to_g, group=group))
            if sizes[frm]:
                ops.append(dist.P2POp(dist.irecv, out[offs[frm]:offs[frm] + sizes[frm]], frm_g, group=group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            if timeout_s is None:
                w.wait()
            else:
                from datetime import timedelta
                w.wait(timedelta(seconds=float(timeout_s)))
        return out.to(local.device) if via_host else out
    mx = max(sizes)
    pad = src
    if src.shape[0] < mx:
        pad = torch.cat([src, src.new_zeros((mx - src.shape[0],) + tuple(src.shape[1:]))], dim=0)
    out = src.new_empty((world * mx,) + tuple(src.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    if via_host:
        out = out.to(local.device)
    if all(sz == mx for sz in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
