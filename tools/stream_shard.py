"""Multi-GPU partitioning of the hot path: independent audio streams (OBS sources) -> ranks.

The path shards by independent unit (one filter instance = one context = one stream,
/root/reference/src/whisper-utils/whisper-utils.cpp:49-50); there is no data-path collective.  The only
cross-rank traffic is the benchmark's barrier + MAX-reduce of the elapsed time.
"""
from __future__ import annotations

from typing import List


def assign_streams(n_streams: int, world_size: int, rank: int) -> List[int]:
    """stream i -> rank i mod world_size."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    return [i for i in range(n_streams) if i % world_size == rank]


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX-reduce a host scalar over the process group (backend-agnostic: gloo on CPU, nccl=RCCL on GPU)."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank: float, world_size: int, elapsed_max: float) -> float:
    """whole-job units / max-over-ranks time (weak scaling: every rank processes the same amount)."""
    return units_per_rank * world_size / elapsed_max
