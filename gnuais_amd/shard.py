"""Channel sharding over the GPUs of one node.

Channels are independent receivers (reference src/ais.c:141-147 creates one
struct receiver per channel, nothing is shared), so the path shards with NO
data-path collective: GPU g owns the contiguous block [g*N/G, (g+1)*N/G) of the
interleaved channel axis and runs its own batch.  torch.distributed is used only
for the benchmark's barrier and for reducing per-rank timings and counters.
"""
from __future__ import annotations


def shard_range(n_channels: int, world: int, rank: int):
    """[lo, hi) of rank's contiguous channel block; blocks tile [0, n_channels)."""
    assert 0 <= rank < world and n_channels >= 0
    return n_channels * rank // world, n_channels * (rank + 1) // world


def reduce_bench(dist, device, seconds: float, msgs: float, samples: float):
    """max over ranks of the elapsed time, sums of messages and samples."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    s = torch.tensor([msgs, samples], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s[0].item()), float(s[1].item())
