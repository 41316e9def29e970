"""Per-clip sharding across the GPUs of one node: clips are independent units, so the forward
path has NO collective -- rank r simply owns a subset of the clip indices.  Only reporting (timing,
optional gather of logits) touches torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_indices(n_clips: int, rank: int, world: int) -> np.ndarray:
    """Round-robin: clip i -> rank (i mod world).  Keeps class-balanced batch composition balanced per rank."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return np.arange(rank, n_clips, world, dtype=np.int64)


def shard_range(n_clips: int, rank: int, world: int):
    """Contiguous split [lo, hi) with the remainder spread over the first ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    q, r = divmod(n_clips, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def class_block_rows(n_classes: int, per_class: int, rank: int, world: int) -> np.ndarray:
    """Rows of rank `rank` in a class-balanced batch [per_class rows of class 0 | per_class of class 1 | ...]: the SAME contiguous
    range shard_range(per_class, rank, world) of every class block.  Contiguous, so that the files feeding a rank's rows are a
    contiguous run of the files the batch was built from (generators.generator runs the front end only for those); the same
    range in every class, so that every rank holds every class in equal numbers."""
    lo, hi = shard_range(per_class, rank, world)
    return np.concatenate([np.arange(c * per_class + lo, c * per_class + hi, dtype=np.int64) for c in range(n_classes)]) \
        if n_classes > 0 else np.zeros(0, np.int64)


def gather_rows(local, indices, n_total, dist=None):
    """Reassemble per-clip rows computed on each rank into global clip order on every rank
    (reporting only).  `local` is a (n_local, d) tensor for `indices` (global clip ids)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(indices, device=local.device)] = local
        return out
    world = dist.get_world_size()
    n_max = (n_total + world - 1) // world
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = torch.as_tensor(shard_indices(n_total, r, world), device=local.device)
        out[idx] = bufs[r][: idx.numel()]
    return out
