"""Multi-GPU side of the path: image sharding and the single exchange step.

The forward pass shards by image (no cross-image op at inference), one process per GPU.
The only exchange is the gather of per-rank detections before evaluation.  The reference
does it with two all_gathers of pickled uint8 payloads (size exchange, then zero-padded
bytes: ufp/mmdet/apis/test.py:161-191); here it is ONE fixed-capacity all_gather of an
fp32 record [imgs_per_rank, cap + 1, 7] whose last row carries the counts -- no pickle,
no size round trip, latency-bound on xGMI (cap = 1000 rows: 224 KB per rank and step).
The record is written by a kernel of the plan (glsdet_pack_detections, captured into the
hipGraph with the forward pass); `DetectionExchange` owns the preallocated receive buffer,
so a step allocates nothing and copies nothing on the host side.

Image -> rank mapping follows the reference's DistributedSampler order that
collect_results_* un-interleaves (test.py:150-155,186-190): image i lives on rank
i % world at local slot i // world.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(num_images: int, rank: int, world: int) -> List[int]:
    """Global image indices handled by `rank` (round-robin, like DistributedSampler without shuffle)."""
    return list(range(rank, num_images, world))


def ranks_agree(flag: bool, mode: str = "all", device="cpu", group=None) -> bool:
    """One answer for every rank: `mode="all"` -- True only if every rank passed True (a failure anywhere ends a
    collective phase everywhere); `mode="any"` -- True if some rank passed True (a clock-driven loop with collectives
    inside runs the same number of rounds on every rank).  A rank must never decide on its own whether to enter a
    collective: the others would wait in it.  No process group -> the local answer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN if mode == "all" else dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def share_tuning(src: int = 0, group=None) -> int:
    """Every rank adopts rank `src`'s kernel-selection table (Engine autotune results, all dtypes), so that all ranks of a
    job run IDENTICAL kernel variants: tuned independently, near-ties fall differently per GPU and the slowest rank's
    choice sets the weak-scaling step.  Call it after `src` has built (tuned) its plans and before the others build
    theirs; ranks != src then find every key in the table and measure nothing.  One object broadcast (a few KB).
    -> number of entries now in the table.  No process group / one rank: a no-op."""
    from . import engine
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sum(len(v) for v in engine._SHARED_TUNED.values())
    rank = dist.get_rank(group)
    box = [{dt: [(list(k), v) for k, v in tab.items()] for dt, tab in engine._SHARED_TUNED.items()}] if rank == src else [None]
    dist.broadcast_object_list(box, src=src, group=group)
    if rank != src:
        for dt, items in box[0].items():
            tab = engine._SHARED_TUNED.setdefault(dt, {})
            for k, v in items:
                tab[tuple(k)] = v
    return sum(len(v) for v in engine._SHARED_TUNED.values())


def pack_detections(dets: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    """dets [n, max_det, 7] + count [>=n] int32 -> [n, max_det+1, 7] fp32 (count in [:, -1, 0])."""
    n, k, f = dets.shape
    out = torch.zeros(n, k + 1, f, dtype=torch.float32, device=dets.device)
    out[:, :k] = dets
    out[:, k, 0] = count[:n].to(torch.float32)
    return out


def gather_detections(dets: torch.Tensor, count: torch.Tensor, group=None) -> torch.Tensor:
    """One collective: every rank gets [world, n, max_det+1, 7]."""
    packed = pack_detections(dets, count)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return packed[None]
    n = packed.shape[0]
    out = torch.empty((world * n,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)       # rank-major concatenation
    return out.view((world, n) + tuple(packed.shape[1:]))


class DetectionExchange:
    """Per plan instance: the preallocated receive buffer of the one all_gather.  `packed` is the
    [n, cap+1, 7] record the plan's glsdet_pack_detections op writes (Engine.pack_detections)."""

    def __init__(self, packed: torch.Tensor, group=None):
        self.packed, self.group = packed, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = packed.shape[0]
        self.out = packed.new_empty((self.world * n,) + tuple(packed.shape[1:])) if self.world > 1 else None

    def gather(self) -> torch.Tensor:
        """-> [world, n, cap+1, 7] on every rank; enqueued on the current stream (call it under the
        stream the plan was launched on, after the launch)."""
        if self.world == 1:
            return self.packed[None]
        dist.all_gather_into_tensor(self.out, self.packed, group=self.group)      # rank-major concatenation
        return self.out.view((self.world,) + tuple(self.packed.shape))


def unpack_in_dataset_order(gathered: torch.Tensor, num_images: Optional[int] = None) -> List[np.ndarray]:
    """[world, n, max_det+1, 7] -> list over GLOBAL image index of ndarray(k, 7), undoing the
    round-robin sharding (image i = rank i % world, slot i // world); padding images
    (beyond num_images) are dropped like the reference's `ordered_results[:size]`."""
    g = gathered.cpu().numpy()
    world, n, k1, _ = g.shape
    total = world * n if num_images is None else num_images
    out = []
    for i in range(total):
        r, slot = i % world, i // world
        cnt = int(round(float(g[r, slot, k1 - 1, 0])))
        out.append(g[r, slot, :cnt].copy())
    return out
