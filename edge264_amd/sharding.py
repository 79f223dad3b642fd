"""Multi-GPU layout of the reconstruction back end: streams are independent objects (a decoder
instance never reads another one's DPB), so N GPUs = N processes, each owning a contiguous shard
of the streams.  There is NO collective on the data path; torch.distributed is only used for the
barrier around the timed region and for the max-over-ranks of the elapsed time (bench.py).
"""
from __future__ import annotations

import os


def rank_info() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_streams(n_streams: int, rank: int, world: int) -> range:
    """Stream ids owned by `rank` when `n_streams` are spread over `world` ranks (contiguous, sizes differ by <= 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def reduce_elapsed(elapsed: float, frames: int, dist=None, device=None) -> tuple[float, int]:
    """max over ranks of the elapsed time, sum over ranks of the frames processed."""
    if dist is None or not dist.is_initialized():
        return elapsed, frames
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    n = torch.tensor([frames], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
