"""Sharding of a block stream over ranks (one process per GPU).

Blocks are independent (integer_group.h:44-60 sees only its own values), so a
large array is cut into contiguous block ranges, one per rank; each rank
encodes its range locally.  The only exchange is one byte count per rank,
which turns shard-local offsets into offsets of the concatenated stream.
Backend-agnostic: works on RCCL ("nccl") and on gloo.
"""
import torch
import torch.distributed as dist

BLOCK_VALUES = 256


def shard_range(n, rank, world, block=BLOCK_VALUES):
    """[lo, hi) in values of rank's contiguous range; boundaries are block multiples."""
    nblocks = (n + block - 1) // block
    per = (nblocks + world - 1) // world
    lo = min(rank * per, nblocks) * block
    hi = min((rank + 1) * per, nblocks) * block
    return min(lo, n), min(hi, n)


def gather_byte_counts(total_bytes, device, group=None):
    """All ranks' stream lengths, in rank order (a list of ints)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(total_bytes)]
    world = dist.get_world_size(group)
    mine = torch.tensor([int(total_bytes)], dtype=torch.int64, device=device)
    out = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def shard_base(counts, rank, align=16):
    """Byte offset of rank's shard in the concatenated stream."""
    base = 0
    for c in counts[:rank]:
        base += (c + align - 1) // align * align
    return base


def max_over_ranks(seconds, device, group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
