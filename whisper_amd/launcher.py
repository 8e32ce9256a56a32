"""Multi-GPU layer: one process per GPU, independent 30 s clips sharded across ranks, ONE collective at load.

The reference has no distributed code at all; the path shards naturally because
DecodingTask.run treats batch rows independently (whisper/decoding.py:713-789) while windows of one file are
sequential (whisper/transcribe.py:272-293).  So: contiguous clip ranges per rank, the packed weight blob is
broadcast once from rank 0 (RCCL over xGMI when the backend is "nccl"), results are gathered as Python
objects.  Nothing is exchanged inside the decode step.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of items owned by `rank`: ceil(n/world) per rank, trailing ranks may be short."""
    per = (n_items + world - 1) // world
    begin = min(rank * per, n_items)
    return begin, min(begin + per, n_items)


def broadcast_weights(blob: Optional[torch.Tensor], dims, dtype: int, device, dist=None) -> torch.Tensor:
    """Rank 0 passes the packed blob (hip.pack_weights); other ranks pass None and receive it.
    Every rank derives the same byte layout from `dims`, so only raw bytes travel."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert blob is not None
        return blob
    from .hip import blob_layout
    _, total = blob_layout(dims, dtype)
    if dist.get_rank() == 0:
        assert blob is not None and blob.numel() == total
    else:
        blob = torch.empty(total, dtype=torch.uint8, device=device)
    dist.broadcast(blob, src=0)
    return blob


def run_sharded(items: Sequence[Any], fn: Callable[[Sequence[Any]], List[Any]], dist=None) -> Optional[List[Any]]:
    """Apply `fn` to this rank's contiguous shard of `items`; rank 0 returns the results of all ranks in the
    original order, other ranks return None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(fn(items))
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = shard_range(len(items), rank, world)
    mine = list(fn(items[b:e])) if e > b else []
    gathered: List[Optional[List[Any]]] = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    out: List[Any] = []
    for part in gathered:
        out.extend(part or [])
    return out
