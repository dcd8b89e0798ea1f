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


# ---- one whole-sequence bitpfr code split over ranks ---------------------------------
#
# bitpfr over a whole array (bitpfr.h:95-124) is one run of 16-byte words of
# cap = 128 // nbits values each, plus two lists with one entry per value that
# does not fit: the gap to the previous such value (offset.h:55-60) and the
# high bits.  Cut at multiples of cap, the packed words of the shards
# concatenate to the words of the whole array; the value lists concatenate as
# they are; of the gap lists only the first entry of each shard changes: it was
# counted from the shard's first value and has to count from the last
# exception of the shards before it.  What the ranks exchange is two numbers
# each: how many exceptions they found and where the last one sits.


def whole_array_range(n, rank, world, nbits):
    """[lo, hi) of rank's range of one packed sequence; boundaries are multiples of
    the values per 16-byte word, so every shard starts a fresh word."""
    cap = 128 // nbits
    nwords = (n + cap - 1) // cap
    per = (nwords + world - 1) // world
    lo = min(rank * per, nwords) * cap
    hi = min((rank + 1) * per, nwords) * cap
    return min(lo, n), min(hi, n)


def gather_exception_summaries(count, last_position, device, group=None):
    """[(exceptions, global position of the last one or -1)] of all ranks, in rank order."""
    if not (dist.is_available() and dist.is_initialized()):
        return [(int(count), int(last_position))]
    world = dist.get_world_size(group)
    mine = torch.tensor([int(count), int(last_position)], dtype=torch.int64, device=device)
    out = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [(int(t[0].item()), int(t[1].item())) for t in out]


def first_gap_fixup(summaries, rank, lo):
    """What to add to the first gap of rank's list: the distance from the value after
    the last exception of the earlier shards (value 0 if there is none) to lo."""
    after = 0
    for count, last in summaries[:rank]:
        if count:
            after = last + 1
    return lo - after


def exception_base(summaries, rank):
    """Index of rank's first exception in the concatenated lists."""
    return sum(count for count, _ in summaries[:rank])


def bitpfr_encode_shard(x_local, lo, origin, nbits, group=None):
    """This rank's part of one whole-sequence bitpfr code on its GPU.

    x_local: the rank's values [lo, lo + len) of the array as a CUDA tensor, lo from
    whole_array_range.  Returns (packed words, gaps, high parts, index of the rank's
    first exception in the concatenated lists): the shards' results put end to end, in
    rank order, are what bitpfr_codec<T>::encode (bitpfr.h:95-99) makes of the whole
    array.  One exchange of two numbers per rank, after the kernels.
    """
    from . import capi

    packed, gaps, highs, count = capi.bitpfr_encode(x_local, origin, nbits)
    last = -1
    if count:
        # position of my last exception: running sum of gap + 1 (offset.h:62-67)
        last = int(lo) + int((gaps.to(torch.int64) + 1).sum().item()) - 1
    summaries = gather_exception_summaries(count, last, x_local.device, group)
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    if count:
        gaps = gaps.clone()
        gaps[0] += first_gap_fixup(summaries, rank, int(lo))
    return packed, gaps, highs, exception_base(summaries, rank)
