"""Multi-GPU partitioning of the (seed x shuffle x game) space and the single tally reduction.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` in CPU tests).  Shuffles are
independent (``docs/rng_contract.md:3-8``), so ranks take contiguous WHOLE deterministic batches and the only
exchange is one integer SUM of the ``[n_batches][S][26]`` tally — the analogue of ``OutcomeCounter.absorb`` +
``_reduce_metric_chunk_payloads`` (``run_tournament.py:197-213, 1023-1042``).  Integer sums are order independent,
so the result is bit-identical to a single-process run.
"""
from __future__ import annotations

import numpy as np


def shard_shuffle_range(shuffle_begin: int, shuffle_end: int, rank: int, world_size: int, batch_size: int = 1) -> tuple[int, int]:
    """Contiguous range of whole batches for ``rank``; batch b = shuffles [b*batch_size, (b+1)*batch_size)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("rank must be in [0, world_size)")
    if batch_size < 1 or shuffle_end < shuffle_begin:
        raise ValueError("bad shuffle range / batch size")
    if shuffle_begin % batch_size:
        raise ValueError("shuffle_begin must be aligned to the deterministic batch size")
    n_batches = -(-(shuffle_end - shuffle_begin) // batch_size)
    lo = (n_batches * rank) // world_size
    hi = (n_batches * (rank + 1)) // world_size
    return (min(shuffle_begin + lo * batch_size, shuffle_end), min(shuffle_begin + hi * batch_size, shuffle_end))


def reduce_tally(tally: np.ndarray, dst: int = 0, device=None) -> np.ndarray:
    """SUM-reduce an int64 tally over the default process group; returns the total on ``dst`` (own tally elsewhere)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tally
    t = torch.from_numpy(np.ascontiguousarray(tally, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
