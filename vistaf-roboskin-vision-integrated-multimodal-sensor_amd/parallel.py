"""Data-parallel sharding of a frame batch over the GPUs of one node (one process per GPU).

Frames are independent given the replicated reference-frame state, so the batch
dimension is split contiguously, every rank runs the whole path on its shard, and the only exchange
is ONE all-gather of the outputs over RCCL/xGMI (`torch.distributed` backend "nccl"), or gloo in the CPU
tests: `PackedGather` packs the requested outputs of a shard (float32 maps, float64 scalar records, ...) into one
byte record per frame, so that a step costs exactly one collective whatever is gathered.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

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

    Shard sizes.  `all_gather_into_tensor` needs equally sized contributions.  Without `total` every rank must hold the same number
    of frames: this is CHECKED once at construction (one tiny setup-time collective) and a ValueError is raised on every rank if
    not.  With `total` (the global frame count, split by `shard_range`) a short last shard is legal: every rank stages
    ceil(total / world) records, the unused tail stays zero, `views()` returns the padded [world * b_max, ...] tensors and
    `compact()` the `total` real frames in global order (one index_select per field).
    """

    def __init__(self, example: Dict[str, torch.Tensor], keys=("height_map_mm", "scalars"), group=None, total: Optional[int] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.keys = tuple(keys)
        b_local = int(example[self.keys[0]].shape[0])
        dev = example[self.keys[0]].device
        if total is None:
            sizes = [b_local] * self.world
            if self.world > 1:
                mine = torch.tensor([b_local], dtype=torch.int64, device=dev)
                got = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(got, mine, group=group)
                sizes = [int(g.item()) for g in got]
                if len(set(sizes)) != 1:
                    raise ValueError(f"PackedGather: shards differ in size {sizes}; pass total= to pad a short last shard")
        else:
            sizes = [b - a for a, b in (shard_range(int(total), r, self.world) for r in range(self.world))]
            if sizes[self.rank] != b_local:
                raise ValueError(f"PackedGather: rank {self.rank} holds {b_local} frames, shard_range({total}, {self.rank}, {self.world}) says {sizes[self.rank]}")
        self.sizes = sizes
        self.b_local = b_local
        b = max(sizes)
        self.fields = []
        off = 0
        for k in self.keys:
            t = example[k]
            if int(t.shape[0]) != b_local:
                raise ValueError("all gathered outputs must have the same number of frames")
            nbytes = t[0].numel() * t.element_size()
            self.fields.append((k, off, nbytes, t.dtype, tuple(t.shape[1:])))
            off += (nbytes + 7) & ~7
        self.frame_bytes = off
        self.b = b
        self.stage = torch.zeros((b, off), dtype=torch.uint8, device=dev)
        self.full = torch.zeros((self.world * b, off), dtype=torch.uint8, device=dev)
        self.padded = any(sz != b for sz in sizes)
        rows = [r * b + i for r, sz in enumerate(sizes) for i in range(sz)]
        self.rows = torch.tensor(rows, dtype=torch.int64, device=dev) if self.padded else None

    def gather(self, local: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for k, off, nbytes, _, _ in self.fields:
            t = local[k]
            if int(t.shape[0]) != self.b_local:
                raise ValueError(f"PackedGather.gather: {k} has {int(t.shape[0])} frames, built for {self.b_local}")
            self.stage[:self.b_local, off:off + nbytes].copy_(t.contiguous().view(self.b_local, -1).view(torch.uint8))
        if self.world == 1:
            self.full.copy_(self.stage)
        else:
            dist.all_gather_into_tensor(self.full, self.stage, group=self.group)
        return self.views()

    def views(self) -> Dict[str, torch.Tensor]:
        """Typed [world * b, ...] views of the gathered records (strided; `.contiguous()` them if needed).  With a short last
        shard the records of rank r start at row r * b and the rows beyond its shard are zero: use `compact()`."""
        out = {}
        for k, off, nbytes, dtype, shape in self.fields:
            esz = torch.empty((), dtype=dtype).element_size()
            flat = self.full.view(dtype)                                       # [n, frame_bytes / esz]: frame_bytes and offsets are multiples of 8
            out[k] = flat[:, off // esz:(off + nbytes) // esz].unflatten(1, shape) if shape else flat[:, off // esz]
        return out

    def compact(self) -> Dict[str, torch.Tensor]:
        """The sum(sizes) real frames in global order (copies; identical to `views()` when no shard is short)."""
        v = self.views()
        return v if not self.padded else {k: t.index_select(0, self.rows) for k, t in v.items()}
