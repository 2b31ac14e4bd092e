"""Multi-GPU layout of the path: only (T) shards.

Each rank owns a contiguous slice of the (i<=j<=k) triple list; the four partial scalars E[T], E(T), D[T], D(T) are
combined with ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests).  CCSD itself runs as replicas.
"""
from __future__ import annotations

import numpy as np


def shard_range(ntriples: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, exhaustive, non-overlapping: rank r gets [r*n//w, (r+1)*n//w).  Equal counts; the spin-free (T) on the
    GPU uses Engine.shard_bounds (afesp_ccsd_t_shard_bounds) instead, which equalises estimated device time."""
    return rank * ntriples // world, (rank + 1) * ntriples // world


def allreduce_scalars(partial: np.ndarray, device=None) -> np.ndarray:
    """Sum the per-rank (T) partials over the default process group; identity when not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(partial, dtype=np.float64)
    t = torch.as_tensor(np.asarray(partial, dtype=np.float64), device=device if device is not None else "cpu")
    dist.all_reduce(t)
    return t.cpu().numpy()
