"""Helpers for the multi-GPU layout of the path: equal-count shard ranges.

Each rank owns a contiguous slice of the triple list -- (i<=j<=k) for the spin-free (T), where the engine's own cost-balanced
bounds (Engine.shard_bounds) are what bench.py and els_amd use, (i<j<k) for the spin-orbital one, which uses shard_range below --
and the partial scalars are combined with ONE all-reduce: the product's afesp_allreduce_sum (ncclAllReduce on the engine stream, or
the host segment when rehearsing ranks share a GPU); gloo only in the CPU tests.  The CCSD iteration runs as replicas unless the
caller opts into the rank split (Engine.ccsd_set_split / AFESP_CC_SHARD=1): bench.py does after checking it on its ranks.
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
