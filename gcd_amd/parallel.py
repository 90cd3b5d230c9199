"""Multi-GPU plumbing for the denoising path: one process per GPU, independent clips per rank.

The path shards naturally: clips are independent, the reference runs one model
replica per GPU with strided example assignment and no communication (scripts/test.py:1059-1084).
The only exchange is collecting the per-rank results: one all-gather (RCCL over xGMI with the
`nccl` backend on ROCm, gloo on CPU for tests) of the final latents — 2 MB per clip.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def clips_for_rank(num_clips: int, rank: int, world: int) -> List[int]:
    """Strided assignment, clip i -> rank i % world (as scripts/test.py:1066-1070 strides examples)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    return list(range(rank, num_clips, world))


def gather_clips(local: torch.Tensor, dist=None, group=None) -> torch.Tensor:
    """all_gather of equally-shaped per-rank results -> [world, *local.shape] on every rank.
    `dist` is torch.distributed (or None for a single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local.unsqueeze(0)
    world = dist.get_world_size(group)
    flat = local.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat, group=group)     # one ring all-gather (RCCL / gloo)
    return out.reshape((world,) + tuple(local.shape))


def gather_ragged_clips(local: List[torch.Tensor], num_clips: int, dist=None,
                        group=None) -> List[Optional[torch.Tensor]]:
    """Collect results of a strided clip assignment back into clip order.  Ranks may hold different
    numbers of clips (num_clips % world != 0): each rank pads its list to the maximum with zeros."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per_rank = (num_clips + world - 1) // world
    assert len(local) == len(clips_for_rank(num_clips, rank, world))
    if per_rank == 0:
        return []
    proto = local[0] if local else None
    shape = torch.tensor(list(proto.shape) if proto is not None else [0],
                         dtype=torch.int64, device="cpu")
    # every rank with >= 1 clip has the same clip shape; rank 0 always has one when num_clips > 0
    shapes = [None] * world
    dist.all_gather_object(shapes, shape.tolist(), group=group)
    clip_shape = next(s for s in shapes if s != [0])
    ref = proto if proto is not None else None
    dev = ref.device if ref is not None else torch.device("cpu")
    dtype = ref.dtype if ref is not None else torch.float32
    stacked = torch.zeros((per_rank,) + tuple(clip_shape), dtype=dtype, device=dev)
    for i, t in enumerate(local):
        stacked[i] = t
    allr = gather_clips(stacked, dist, group)               # [world, per_rank, ...]
    out: List[Optional[torch.Tensor]] = [None] * num_clips
    for r in range(world):
        for j, clip in enumerate(clips_for_rank(num_clips, r, world)):
            out[clip] = allr[r, j]
    return out
