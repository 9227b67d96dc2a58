"""Data-parallel sharding of a frame batch over the GPUs of one node (one process per GPU).

Frames are independent given the replicated reference-frame state, so the batch
dimension is split contiguously, every rank runs the whole path on its shard, and the only exchange
is ONE all-gather of the outputs (height maps, or the [B,16] scalar records) over RCCL/xGMI
(`torch.distributed` backend "nccl"), or gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `total` frames owned by `rank`; the first total % world ranks get one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_outputs(local: Dict[str, torch.Tensor], group=None, keys=("height_map_mm", "scalars")) -> Dict[str, torch.Tensor]:
    """One all-gather per requested output.  Shards must be equally sized (pad the last shard upstream);
    the default gathers the [B/W,h,w] float32 maps and the [B/W,16] float64 scalar records."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: local[k] for k in keys}
    world = dist.get_world_size(group)
    out = {}
    for k in keys:
        t = local[k].contiguous()
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(full, t, group=group)
        out[k] = full
    return out
