"""Data-parallel sharding of a frame batch over the GPUs of one node (one process per GPU).

Frames are independent given the replicated reference-frame state, so the batch
dimension is split contiguously, every rank runs the whole path on its shard, and the only exchange
is ONE all-gather of the outputs over RCCL/xGMI (`torch.distributed` backend "nccl"), or gloo in the CPU
tests: `PackedGather` packs the requested outputs of a shard (float32 maps, float64 scalar records, ...) into one
byte record per frame, so that a step costs exactly one collective whatever is gathered.
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


class PackedGather:
    """ONE all-gather per step for several per-frame outputs.

    The outputs of a shard ([b, ...] tensors of any dtype) are copied side by side into a [b, bytes_per_frame] uint8
    staging buffer, gathered with a single `all_gather_into_tensor`, and exposed as typed views of the gathered buffer
    (no unpacking copy: every field starts at an 8-byte aligned offset of the per-frame record).  Buffers are allocated
    once and reused, so the collective can run on its own stream while the next batch is being computed.
    """

    def __init__(self, example: Dict[str, torch.Tensor], keys=("height_map_mm", "scalars"), group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.keys = tuple(keys)
        b = int(example[self.keys[0]].shape[0])
        dev = example[self.keys[0]].device
        self.fields = []
        off = 0
        for k in self.keys:
            t = example[k]
            if int(t.shape[0]) != b:
                raise ValueError("all gathered outputs must have the same number of frames")
            nbytes = t[0].numel() * t.element_size()
            self.fields.append((k, off, nbytes, t.dtype, tuple(t.shape[1:])))
            off += (nbytes + 7) & ~7
        self.frame_bytes = off
        self.b = b
        self.stage = torch.empty((b, off), dtype=torch.uint8, device=dev)
        self.full = torch.empty((self.world * b, off), dtype=torch.uint8, device=dev)

    def gather(self, local: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for k, off, nbytes, _, _ in self.fields:
            self.stage[:, off:off + nbytes].copy_(local[k].contiguous().view(self.b, -1).view(torch.uint8))
        if self.world == 1:
            self.full.copy_(self.stage)
        else:
            dist.all_gather_into_tensor(self.full, self.stage, group=self.group)
        return self.views()

    def views(self) -> Dict[str, torch.Tensor]:
        """Typed [world * b, ...] views of the gathered records (strided; `.contiguous()` them if needed)."""
        out = {}
        n = self.world * self.b
        for k, off, nbytes, dtype, shape in self.fields:
            esz = torch.empty((), dtype=dtype).element_size()
            flat = self.full.view(dtype)                                       # [n, frame_bytes / esz]: frame_bytes and offsets are multiples of 8
            out[k] = flat[:, off // esz:(off + nbytes) // esz].unflatten(1, shape) if shape else flat[:, off // esz]
        return out
