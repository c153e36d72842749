"""Block-range sharding across ranks: rank r owns the contiguous
block range [r*B/G, (r+1)*B/G); output order = block order.  The only exchange on
the path is one all_gather of the per-rank compressed totals, from which every
rank derives the base offset of its span in the concatenated stream.  Payload
never moves between GPUs."""


def block_range(nblocks, rank, world):
    lo = nblocks * rank // world
    hi = nblocks * (rank + 1) // world
    return lo, hi


def bases_from_totals(totals):
    """exclusive prefix sum of the all-gathered per-rank totals"""
    out, run = [], 0
    for t in totals:
        out.append(run)
        run += int(t)
    return out, run


def exchange_totals(local_total, device=None):
    """all_gather of one int64 per rank (RCCL over xGMI with backend nccl, gloo on
    CPU).  Returns the python list of totals in rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    allt = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allt, mine)
    return [int(x) for x in allt.cpu()]
