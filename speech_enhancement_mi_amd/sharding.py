"""Multi-GPU sharding of the streaming path: streams are independent units (every norm and state is
per stream), so rank r simply owns a contiguous slice of the B streams with a full weight replica.  No data-path
collective exists; the only collectives are the barrier and the MAX-over-ranks of the timed region in bench.py."""
from __future__ import annotations


def shard_streams(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` streams for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device="cuda") -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
