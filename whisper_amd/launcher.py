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


def balanced_shards(costs: Sequence[float], world: int) -> List[List[int]]:
    """Item indices per rank, longest-processing-time first: items are taken in descending cost and given to the
    rank with the least work so far (ties: lowest rank).  Files of very different lengths would otherwise leave the
    ranks with contiguous shards idle while the longest shard finishes (load imbalance is the only
    limit on scaling — nothing is exchanged in the step).  Each rank's list is returned in ascending index order."""
    loads = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i)):
        r = min(range(world), key=lambda r: (loads[r], r))
        shards[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(s) for s in shards]


def run_balanced(items: Sequence[Any], costs: Sequence[float], fn: Callable[[Sequence[Any]], List[Any]],
                 dist=None) -> Optional[List[Any]]:
    """`run_sharded` with cost-balanced instead of contiguous shards; rank 0 returns all results in input order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(fn(items))
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = balanced_shards(costs, world)
    mine = list(fn([items[i] for i in shards[rank]])) if shards[rank] else []
    if len(mine) != len(shards[rank]):
        raise RuntimeError(f"rank {rank}: {len(mine)} results for {len(shards[rank])} items")
    gathered: List[Optional[List[Any]]] = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    out: List[Any] = [None] * len(items)
    for idx, part in zip(shards, gathered):
        for i, res in zip(idx, part or []):
            out[i] = res
    return out


def _audio_cost(audio) -> float:
    """work estimate of one input of transcribe(): samples of an array, bytes of a file"""
    import os
    if isinstance(audio, (str, bytes, os.PathLike)):
        try:
            return float(os.path.getsize(audio))
        except OSError:
            return 1.0
    return float(getattr(audio, "shape", [1])[-1])


def transcribe_sharded(model, audios: Sequence[Any], dist=None, *, batch_size: int = 24, **kwargs) -> Optional[List[dict]]:
    """Many files over many GPUs: every rank transcribes its cost-balanced share with `transcribe_batch` (windows of
    one file never leave their rank: seek and prompt depend on the previous window, transcribe.py:288-293,371-399);
    rank 0 returns the result dicts in input order, other ranks None.  All inputs must be of one kind (arrays or
    paths) for the costs to be comparable."""
    from .transcribe import transcribe_batch
    costs = [_audio_cost(a) for a in audios]
    return run_balanced(list(audios), costs, lambda part: transcribe_batch(model, part, batch_size=batch_size, **kwargs),
                        dist)
