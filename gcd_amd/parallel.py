"""Multi-GPU plumbing for the denoising path: one process per GPU, independent clips per rank.

The path shards naturally: clips are independent, the reference runs one model
replica per GPU with strided example assignment and no communication (scripts/test.py:1059-1084).
The only exchange is collecting the per-rank results: one all-gather (RCCL over xGMI with the
`nccl` backend on ROCm, gloo on CPU for tests) of the final latents — 2 MB per clip.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

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
    # exchange shape, dtype and device TYPE: a rank that holds no clip must still contribute a pad
    # tensor of the same dtype on the same kind of device as everybody else, or the all-gather
    # errors (gloo) / hangs (RCCL)
    meta = None if proto is None else (tuple(proto.shape), str(proto.dtype).replace("torch.", ""),
                                       proto.device.type)
    metas = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    clip_shape, dtype_name, dev_type = next(m for m in metas if m is not None)
    assert all(m is None or tuple(m) == (clip_shape, dtype_name, dev_type) for m in metas), \
        f"ranks disagree on the clip shape / dtype / device: {metas}"
    dtype = getattr(torch, dtype_name)
    if proto is not None:
        dev = proto.device
    elif dev_type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device(dev_type)
    stacked = torch.zeros((per_rank,) + tuple(clip_shape), dtype=dtype, device=dev)
    for i, t in enumerate(local):
        stacked[i] = t
    allr = gather_clips(stacked, dist, group)               # [world, per_rank, ...]
    out: List[Optional[torch.Tensor]] = [None] * num_clips
    for r in range(world):
        for j, clip in enumerate(clips_for_rank(num_clips, r, world)):
            out[clip] = allr[r, j]
    return out


def sample_clips(sample_one: Callable[[int, torch.Generator], torch.Tensor], num_clips: int,
                 dist=None, group=None, base_seed: int = 0, device=None,
                 gather: bool = True) -> List[Optional[torch.Tensor]]:
    """Sharded-clip driver: the multi-GPU shape of the reference's evaluation loop
    (scripts/test.py:1051-1090 — one process + one model replica per GPU, examples strided over the
    workers, no communication while sampling).

    Rank r runs `sample_one(clip_index, generator)` for clips r, r + world, r + 2 world, ...; the
    generator is seeded by `base_seed + clip_index`, so a clip's noise — and therefore its result —
    does not depend on how many GPUs the job runs on.  With `gather`, one all-gather (RCCL over xGMI
    / gloo) after the loop returns every clip's result on every rank, in clip order; otherwise each
    rank gets its own clips in place and None elsewhere.

    `sample_one` is typically
        lambda i, g: sampler(denoiser_i, torch.randn(shape, generator=g, device=dev), cond_i, uc_i)
    """
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size(group)
    rank = 0 if world == 1 else dist.get_rank(group)
    mine = clips_for_rank(num_clips, rank, world)
    local = []
    for i in mine:
        g = torch.Generator(device=device) if device is not None else torch.Generator()
        g.manual_seed(base_seed + i)
        local.append(sample_one(i, g))
    if not gather:
        out: List[Optional[torch.Tensor]] = [None] * num_clips
        for i, t in zip(mine, local):
            out[i] = t
        return out
    return gather_ragged_clips(local, num_clips, dist, group)
